"""GPU probe (not a test): why bench.py's ms_per_step at the headline is above the phase probe's solve_fermion median.
Times solve_fermion in the bench's own setting: torch's HIP context alive, event sampling on, 20-step windows."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S, fermion as F

h1, eri = S.synthetic_integrals(30)
sa, sb = S.uniform_strings(30, 8, 317, 1000), S.uniform_strings(30, 8, 317, 1000 + 7919)


def window(n=20, stats=False):
    t0 = time.perf_counter()
    for _ in range(n):
        F.solve_fermion((sa, sb), h1, eri)
        if stats:
            F.last_solve_stats()
    return (time.perf_counter() - t0) / n * 1e3


def report(tag):
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        F.solve_fermion((sa, sb), h1, eri)
    w = [window() for _ in range(15)]
    print(f'{tag:46s} 20-step windows: min {min(w):.3f} med {np.median(w):.3f} max {max(w):.3f} ms/step', flush=True)


report('no torch, no events')
F.set_profiling(8)
report('no torch, events every 8')
F.set_profiling(0)
import torch
torch.cuda.synchronize(torch.device('cuda', 0))
report('torch context alive, no events')
F.set_profiling(8)
report('torch context alive, events every 8')
w = [window(stats=True) for _ in range(15)]
print(f'{"... + last_solve_stats per step":46s} 20-step windows: min {min(w):.3f} med {np.median(w):.3f} max {max(w):.3f}')
def synced_window():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        F.solve_fermion((sa, sb), h1, eri); F.last_solve_stats()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
w = [synced_window() for _ in range(15)]
print(f'{"... + torch.cuda.synchronize around":46s} 20-step windows: min {min(w):.3f} med {np.median(w):.3f} max {max(w):.3f}')
F.set_profiling(0)
