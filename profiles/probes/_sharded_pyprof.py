"""GPU probe (not a test): cProfile of solve_sci_sharded on an RCCL group of one (HF-centred 317 x 317, sparse work items)."""
import cProfile, io, os, pstats, socket, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import torch.distributed as dist
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import fermion as F
from qiskit_addon_sqd_amd.sharded import solve_sci_sharded

os.environ.setdefault('SQD_SIGMA_DENSE', '0')
h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
sa, sb = S.hf_centred_strings(30, 8, 317, 100), S.hf_centred_strings(30, 8, 317, 900)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for _ in range(5):
    solve_sci_sharded((sa, sb), h1, eri, 30, (8, 8), gather_state=False)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    solve_sci_sharded((sa, sb), h1, eri, 30, (8, 8), gather_state=False)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(32)
print(s.getvalue()[:6000])
dist.destroy_process_group()
