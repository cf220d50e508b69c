import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd.fermion import solve_sci, solve_fermion
h1, eri = S.synthetic_integrals(30)
sa, sb = S.uniform_strings(30, 8, 317, 100), S.uniform_strings(30, 8, 317, 900)
for _ in range(3): solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
t0=time.perf_counter()
for _ in range(20): solve_fermion((sa, sb), h1, eri)
print('solve_fermion ms', (time.perf_counter()-t0)/20*1e3)
