"""GPU probe (not a test): cProfile of the headline solve_fermion call -- where the Python layer's share of a step goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import fermion as F

h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
sa, sb = S.uniform_strings(30, 8, 317, 1000), S.uniform_strings(30, 8, 317, 1000 + 7919)
f = lambda: F.solve_fermion((sa, sb), h1, eri, spin_sq=None)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    f()
ts = []
for _ in range(300):
    t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
print(f"solve_fermion: median {np.median(ts) * 1e3:.4f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    f()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
