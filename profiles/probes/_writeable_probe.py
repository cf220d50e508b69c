"""GPU probe (not a test): solve_fermion with frozen (read-only) against plain writeable integral tensors, uniform 317^2."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S, fermion as F
h1w, eriw = S.synthetic_integrals(30)
h1w, eriw = np.array(h1w), np.array(eriw)
h1, eri = F.freeze_integrals(h1w, eriw)
sa, sb = S.uniform_strings(30, 8, 317, 1000), S.uniform_strings(30, 8, 317, 1000 + 7919)
import gc; gc.collect(); gc.freeze()
def run(h, e, n):
    t0 = time.perf_counter()
    for _ in range(n):
        F.solve_fermion((sa, sb), h, e)
    return 1e3 * (time.perf_counter() - t0) / n
run(h1, eri, 1500); run(h1w, eriw, 200)
for rep in range(3):
    print(f'frozen {run(h1, eri, 300):.4f} ms   writeable {run(h1w, eriw, 300):.4f} ms', flush=True)
