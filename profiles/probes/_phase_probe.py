"""GPU probe (not a test): wall-clock per phase of one native solve at the headline size."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
for name, gen in (('uniform', S.uniform_strings), ('hf', S.hf_centred_strings)):
    sa, sb = gen(30, 8, 317, 1000), gen(30, 8, 317, 1000 + 7919)
    ctx = _capi.Context(h1, eri)
    for _ in range(3):
        ctx.set_subspace(sa, sb); ctx.davidson(); ctx.observables()
    acc = np.zeros(5); n = 20
    for _ in range(n):
        t0 = time.perf_counter(); ctx.set_subspace(sa, sb)
        t1 = time.perf_counter(); amps, st = ctx.davidson(fetch=False)
        t2 = time.perf_counter(); ctx.observables()
        t3 = time.perf_counter(); amps, st2 = ctx.davidson(fetch=True)
        t4 = time.perf_counter()
        acc += [t1 - t0, t2 - t1, t3 - t2, (t4 - t3) - (t2 - t1), st['ms_total'] * 1e-3]
    print(name, 'ms: set_subspace %.3f davidson(no fetch) %.3f observables %.3f amps_fetch_extra %.3f davidson_device %.3f n_sigma %d' % (*(acc / n * 1e3), st['n_sigma']))
    ctx.close()
