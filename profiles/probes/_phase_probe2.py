"""GPU probe (not a test): where the wall time of one solve goes, native phases and the Python API, both generators."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S, fermion as F, _capi

h1, eri = S.synthetic_integrals(30)
ctx = F._get_context(h1, eri, 0)


def t(f, n=30):
    """median wall time of one call, ms (the median: a process' first ~0.1 s of GPU activity contains one or two
    30-50 ms stalls, stall_probe.txt, which a mean over 30 calls would charge to whichever phase they land in)"""
    for _ in range(3):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


for name, gen in (('uniform', S.uniform_strings), ('hf', S.hf_centred_strings)):
    sa, sb = gen(30, 8, 317, 1000), gen(30, 8, 317, 1000 + 7919)
    n = 30 if name == 'uniform' else 8
    r = {}
    r['set_subspace+sync'] = t(lambda: (ctx.set_subspace(sa, sb), ctx.hdiag()), n)  # hdiag = drain + 0.8 MB copy
    ctx.set_subspace(sa, sb)
    r['davidson(no fetch)'] = t(lambda: ctx.davidson(fetch=False), n)
    r['davidson(fetch)'] = t(lambda: ctx.davidson(), n)
    r['davidson+observables'] = t(lambda: ctx.davidson(observables=True), n)
    r['native solve (strings)'] = t(lambda: ctx.solve(sa, sb), n)
    r['native solve, no S^2'] = t(lambda: ctx.solve(sa, sb, spin_square=False), n)
    r['solve_fermion API'] = t(lambda: F.solve_fermion((sa, sb), h1, eri), n)
    F.set_profiling(4)
    r['solve_fermion API, events every 4'] = t(lambda: F.solve_fermion((sa, sb), h1, eri), n)
    F.set_profiling(0)
    r['solve_sci API'] = t(lambda: F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False), n)
    st = F.last_solve_stats()
    print(name, 'ms:', ' | '.join(f'{k} {v:.3f}' for k, v in r.items()), '| n_sigma', st['n_sigma'], flush=True)
