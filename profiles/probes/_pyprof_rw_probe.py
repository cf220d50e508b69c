"""GPU probe: the Python layer of a solve_fermion call with plain (writeable) integral tensors (cProfile, 3000 calls), and
the same loop with the hash threads switched off / on (SQD_HASH_THREADS) for the cost of their company."""
import cProfile, pstats, os, sys, time, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S, fermion as F
h1, eri = S.synthetic_integrals(30)
h1, eri = np.array(h1), np.array(eri)
hf, ef = F.freeze_integrals(h1.copy(), eri.copy())
sa, sb = S.uniform_strings(30, 8, 317, 11), S.uniform_strings(30, 8, 317, 13)
for _ in range(300): r = F.solve_fermion((sa, sb), h1, eri); r[1].amplitudes
n = 3000
for name, (a, b) in (("frozen", (hf, ef)), ("writeable", (h1, eri)), ("frozen", (hf, ef)), ("writeable", (h1, eri))):
    for _ in range(100): r = F.solve_fermion((sa, sb), a, b); r[1].amplitudes
    t = time.perf_counter()
    for _ in range(n): r = F.solve_fermion((sa, sb), a, b); x = r[1].amplitudes
    print(f"{name}: {1e3*(time.perf_counter()-t)/n:.4f} ms per call  (SQD_HASH_THREADS={os.environ.get('SQD_HASH_THREADS','default')})")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): r = F.solve_fermion((sa, sb), h1, eri); x = r[1].amplitudes
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(16); print(s.getvalue())
