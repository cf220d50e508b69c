"""Multi-process CPU test (gloo, world_size 2) of the batch-sharded ``sci_solver``: round-robin
ownership, the single all-reduce of (E, occ) records, argmin, winner broadcast, 'mean' option."""
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_indices():
    from qiskit_addon_sqd_amd.distributed import shard_indices

    assert shard_indices(8, 0, 8) == [0] and shard_indices(8, 7, 8) == [7]
    assert shard_indices(5, 1, 2) == [1, 3] and shard_indices(1, 1, 2) == []
    assert sorted(sum((shard_indices(11, r, 4) for r in range(4)), [])) == list(range(11))


def _run_two_ranks(worker: str, world: int = 2, **extra_env):
    from conftest import EMU_LIB

    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SQD_EMU_LIB=str(EMU_LIB), OMP_NUM_THREADS="1", **extra_env)
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / worker)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"rank {rank} ok" in out


def test_two_rank_gloo_batch_sharding(emu_lib):
    _run_two_ranks("_dist_worker.py")


def test_two_rank_gloo_whole_loop_unseeded(emu_lib):
    """The package's own SQD loop in SPMD mode with ``seed=None``: CI strings prepared on rank 0 and broadcast, the
    collective solver in the middle, the iteration state broadcast back (reference fermion.py:421-451)."""
    _run_two_ranks("_dist_loop_worker.py")


import pytest  # noqa: E402


# (SQD_SIGMA_DENSE=0: a row shard always runs the sparse same-spin work items, and the bit-for-bit comparison below is
# against the whole-subspace sigma of the SAME kernels)
@pytest.mark.parametrize("kernel_env", [{"SQD_SIGMA_DENSE": "0"}, {"SQD_SIGMA_ROWS": "2"}, {"SQD_SIGMA_DIRECT": "1"}],
                         ids=["work-items", "rows", "direct"])
def test_two_rank_gloo_row_sharded_sigma_and_solve(emu_lib, kernel_env):
    """one subspace split by alpha rows over two ranks -- all-gather of the vector, sigma rows per rank
    bit-identical to the single-rank sigma, and the collective Davidson against dense diagonalisation; with each of
    the three sigma kernels on the row range."""
    _run_two_ranks("_dist_shard_worker.py", **kernel_env)


def test_one_rank_group_single_call_iteration(emu_lib):
    """A group of ONE rank (what a 1-GPU run of a multi-GPU script is): the row-sharded solver's iteration as one native
    call with the fused dots / eigen kernel, against the staged calls of a real group -- bit for bit -- and against
    dense diagonalisation; with and without the linear spin penalty."""
    _run_two_ranks("_dist_alone_worker.py", world=1)

