"""GPU probe (not a test; run under torchrun with one process): host-side cost of each piece of the per-step exchange of
solve_sci_batch_distributed on an RCCL group, measured around the real calls."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, torch.distributed as dist
from qiskit_addon_sqd_amd import synthetic as S, fermion as F, distributed as D, _capi

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
h1, eri = S.synthetic_integrals(30)
sa, sb = S.uniform_strings(30, 8, 317, 1000), S.uniform_strings(30, 8, 317, 8919)
batches = [(sa, sb)]
def med(f, n=200):
    for _ in range(20): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts))
print('solve_sci alone            %.1f us' % med(lambda: F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)))
print('solve_sci_batch_distributed %.1f us' % med(lambda: D.solve_sci_batch_distributed(batches, h1, eri, 30, (8, 8), compute_rdms=False)))
dt, ht, _xch = D._exchange_buffers(None, dev, 1, 61, True)
def table():
    dt.copy_(ht, non_blocking=True); dist.all_reduce(dt); ht.copy_(dt, non_blocking=True); D._wait_stream(torch.cuda.current_stream())
print('table exchange             %.1f us' % med(table))
def ar_only():
    dist.all_reduce(dt); D._wait_stream(torch.cuda.current_stream())
print('  all_reduce + wait         %.1f us' % med(ar_only))
ta = torch.empty((317, 317), dtype=torch.float64, device=dev)
def bc():
    dist.broadcast(ta, src=0); D._wait_stream(torch.cuda.current_stream())
print('broadcast 0.8 MB + wait    %.1f us' % med(bc))
amps = _capi.pinned_empty((317, 317))
def d2h():
    torch.from_numpy(amps).copy_(ta, non_blocking=True); torch.cuda.current_stream().synchronize()
print('D2H 0.8 MB pinned + sync   %.1f us' % med(d2h))
res = F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)
print('resident view              %.1f us' % med(lambda: D._resident_solution(h1, eri, 0, (317, 317), dev)))
print('SCIResult rebuild          %.1f us' % med(lambda: F.SCIResult(res.energy, res.sci_state, res.orbital_occupancies)))
def sync_blocking():
    dist.all_reduce(dt); torch.cuda.current_stream().synchronize()
print('  all_reduce + blocking sync %.1f us' % med(sync_blocking))
def copies_only():
    dt.copy_(ht, non_blocking=True); ht.copy_(dt, non_blocking=True); D._wait_stream(torch.cuda.current_stream())
print('  two 488-byte copies + wait %.1f us' % med(copies_only))
dist.destroy_process_group()
