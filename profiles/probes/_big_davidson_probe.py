"""GPU probe: a whole Davidson solve at uniform N x N (default 10^4 x 10^4, D = 10^8): time per iteration against the sigma
application alone, under SQD_RED_BLOCKS settings given in the environment."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import synthetic as S, fermion as F, _capi
if os.environ.get('SQD_LIB'):
    from pathlib import Path
    _capi.LIB_PATH = Path(os.environ['SQD_LIB'])
n = int(os.environ.get('N', '10000'))
h1, eri = S.synthetic_integrals(30)
ctx = F._get_context(h1, eri, 0)
sa, sb = S.uniform_strings(30, 8, n, 11), S.uniform_strings(30, 8, n, 13)
ctx.set_subspace(sa, sb); ctx.sync()
ms_sigma = ctx.time_sigma(3)
for rep in range(2):
    t0 = time.perf_counter()
    amps, st = ctx.davidson(fetch=False) if 'fetch' in ctx.davidson.__code__.co_varnames else ctx.davidson()
    ctx.sync()
    dt = time.perf_counter() - t0
print({k: os.environ[k] for k in os.environ if k.startswith('SQD_')}, f"n={n} {ctx.sigma_kernel()} sigma alone {ms_sigma:.3f} ms | davidson {1e3*dt:.1f} ms wall, "
      f"{st['n_sigma']} sigma builds, {st['iterations']} iterations -> {1e3*dt/max(st['n_sigma'],1):.2f} ms per sigma build; converged {st['converged']}")
