"""GPU probe (not a test): Davidson time against the grid of its BLAS-1 kernels (SQD_RED_BLOCKS, read at first use)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S, fermion as F
h1, eri = S.synthetic_integrals(30)
ctx = F._get_context(h1, eri, 0)
out = []
for name, gen, n in (('uniform', S.uniform_strings, 60), ('hf', S.hf_centred_strings, 12)):
    sa, sb = gen(30, 8, 317, 1000), gen(30, 8, 317, 1000 + 7919)
    ctx.set_subspace(sa, sb)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        ctx.davidson(fetch=False)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); _, st = ctx.davidson(fetch=False); ts.append((time.perf_counter() - t0) * 1e3)
    out.append(f'{name} davidson {np.median(ts):.3f} ms ({st["n_sigma"]} sigma, {1e3*np.median(ts)/st["n_sigma"]:.1f} us/iter, eig solves {st["n_eig_solves"]}, jacobi {st["n_eig_fallbacks"]})')
print('SQD_RED_BLOCKS=' + os.environ.get('SQD_RED_BLOCKS', 'default(197)'), ' | '.join(out), flush=True)
