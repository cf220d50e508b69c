"""GPU probe (not a test): per-call wall times of repeated identical solves -- are there sporadic stalls?"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S, fermion as F

h1, eri = S.synthetic_integrals(30)
ctx = F._get_context(h1, eri, 0)
sa, sb = S.uniform_strings(30, 8, 317, 1000), S.uniform_strings(30, 8, 317, 1000 + 7919)


def series(f, n=200):
    for _ in range(5):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    big = np.nonzero(ts > 3 * np.median(ts))[0]
    return f'min {ts.min():.3f} med {np.median(ts):.3f} p90 {np.percentile(ts, 90):.3f} max {ts.max():.3f} mean {ts.mean():.3f} outliers@{big.tolist()[:12]}'


for rep in range(2):
    print('native solve      ', series(lambda: ctx.solve(sa, sb)), flush=True)
    print('native no S^2     ', series(lambda: ctx.solve(sa, sb, spin_square=False)), flush=True)
    print('solve_fermion     ', series(lambda: F.solve_fermion((sa, sb), h1, eri)), flush=True)
    print('solve_sci         ', series(lambda: F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)), flush=True)
    print('set_subspace only ', series(lambda: ctx.set_subspace(sa, sb)), flush=True)
    ctx.set_subspace(sa, sb)
    print('davidson no fetch ', series(lambda: ctx.davidson(fetch=False)), flush=True)
