"""GPU probe: where the Python layer of a uniform 317 x 317 solve_fermion call spends its time (cProfile, 3000 calls)."""
import cProfile, pstats, os, sys, time, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S, fermion as F
h1, eri = S.synthetic_integrals(30)
h1, eri = F.freeze_integrals(h1, eri)
sa, sb = S.uniform_strings(30, 8, 317, 11), S.uniform_strings(30, 8, 317, 13)
for _ in range(300): r = F.solve_fermion((sa, sb), h1, eri); r[1].amplitudes
n = 3000
t = time.perf_counter()
for _ in range(n): r = F.solve_fermion((sa, sb), h1, eri); a = r[1].amplitudes
print(f"unprofiled: {1e3*(time.perf_counter()-t)/n:.4f} ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): r = F.solve_fermion((sa, sb), h1, eri); a = r[1].amplitudes
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue())
