"""GPU probe (not a test): cost of the RDM outputs that solve_sci returns (reference fermion.py:725-742)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import _capi, synthetic as S
from qiskit_addon_sqd_amd.fermion import solve_sci
h1, eri = S.synthetic_integrals(30)
for name, gen in (('uniform', S.uniform_strings), ('hf', S.hf_centred_strings)):
    sa, sb = gen(30, 8, 317, 1000), gen(30, 8, 317, 1000 + 7919)
    ctx = _capi.Context(h1, eri)
    ctx.set_subspace(sa, sb); ctx.davidson(fetch=False)
    for fn in ('rdm1s', 'rdm2'):
        getattr(ctx, fn)()
        t0 = time.perf_counter()
        for _ in range(5): getattr(ctx, fn)()
        print(name, fn, 'ms', (time.perf_counter() - t0) / 5 * 1e3, flush=True)
    ctx.close()
    solve_sci((sa, sb), h1, eri, 30, (8, 8))
    t0 = time.perf_counter()
    for _ in range(5): r = solve_sci((sa, sb), h1, eri, 30, (8, 8))
    print(name, 'solve_sci (with rdm1, rdm2, energy einsum) ms', (time.perf_counter() - t0) / 5 * 1e3, flush=True)
