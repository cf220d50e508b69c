"""GPU probe (not a test): the row-sharded collective solver on an RCCL group of ONE rank against solve_sci on the same
HF-centred 317 x 317 subspace (what the collective machinery costs when there is nothing to exchange)."""
import os, socket, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import torch
import torch.distributed as dist
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import fermion as F
from qiskit_addon_sqd_amd.sharded import solve_sci_sharded

os.environ.setdefault('SQD_SIGMA_DENSE', '0')  # (a row shard runs the sparse same-spin work items: like with like)
h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
sa, sb = S.hf_centred_strings(30, 8, 317, 100), S.hf_centred_strings(30, 8, 317, 900)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))


def timed(fn, reps=7):
    for _ in range(3):
        out = fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), out


t_ref, ref = timed(lambda: F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False))
n_ref = F.last_solve_stats()['n_sigma']
t_sh, res = timed(lambda: solve_sci_sharded((sa, sb), h1, eri, 30, (8, 8), gather_state=False))
st = res._sharded_stats
print(f"solve_sci            : {t_ref:8.3f} ms, {n_ref} sigma builds, E = {ref.energy:.10f}")
os.environ["SQD_SHARD_FORCE_COLLECTIVES"] = "1"
t_f, res_f = timed(lambda: solve_sci_sharded((sa, sb), h1, eri, 30, (8, 8), gather_state=False))
del os.environ["SQD_SHARD_FORCE_COLLECTIVES"]
t_t, res_t = timed(lambda: solve_sci_sharded((sa, sb), h1, eri, 30, (8, 8), gather_state=False, driver="torch"))
print(f"solve_sci_sharded ws=1, collectives really issued (all-gather + 2 all-reduces per iteration on the 1-rank group): {t_f:8.3f} ms (ratio {t_f / t_ref:.2f})")
print(f"solve_sci_sharded ws=1, torch-level driver (round 2 structure, two host reads per iteration): {t_t:8.3f} ms (ratio {t_t / t_ref:.2f})")
print(f"solve_sci_sharded ws=1: {t_sh:8.3f} ms, {st['n_sigma']} sigma builds, E = {res.energy:.10f}  (ratio {t_sh / t_ref:.2f}, "
      f"per iteration {1e3 * t_sh / st['n_sigma']:.1f} us vs {1e3 * t_ref / n_ref:.1f} us)")
dist.destroy_process_group()
