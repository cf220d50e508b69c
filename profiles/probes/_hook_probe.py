"""GPU probe (not a test): per-step wall clock of solve_sci_batch_distributed on an RCCL group of one rank with the exchange
enqueued by the solve's hook (SQD_DIST_HOOK=1, default) or after the solve (=0), and the host time spent inside the hook."""
import os, socket, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import torch
import torch.distributed as dist
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import fermion as F
from qiskit_addon_sqd_amd import distributed as D

h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
sa, sb = S.uniform_strings(30, 8, 317, 1000), S.uniform_strings(30, 8, 317, 1000 + 7919)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
real_all_reduce = dist.all_reduce
hook_t = []
def timed_all_reduce(*a, **k):
    t0 = time.perf_counter(); r = real_all_reduce(*a, **k); hook_t.append((time.perf_counter() - t0) * 1e6); return r
dist.all_reduce = timed_all_reduce
f = lambda: D.solve_sci_batch_distributed([(sa, sb)], h1, eri, 30, (8, 8), compute_rdms=False)
if os.environ.get('PROFILING'):
    F.set_profiling(time_sigma_every=int(os.environ['PROFILING']))  # (what bench.py switches on: event brackets around sampled sigma launches)
for _ in range(10):
    f()
hook_t.clear()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); r = f(); _ = r[0].sci_state.amplitudes[0, 0]; ts.append((time.perf_counter() - t0) * 1e3)
ts, ht = np.array(ts), np.array(hook_t)
print(f"hook {os.environ.get('SQD_DIST_HOOK', '1')}: step ms median {np.median(ts):.3f} mean {ts.mean():.3f} p90 {np.percentile(ts, 90):.3f} max {ts.max():.3f} | "
      f"all_reduce call us median {np.median(ht):.1f} mean {ht.mean():.1f} max {ht.max():.1f} | steps over 0.4 ms: {(ts > 0.4).sum()} at {np.nonzero(ts > 0.4)[0][:12]}")
dist.destroy_process_group()
