"""Worker of tests/test_distributed.py::test_two_rank_gloo_whole_loop_unseeded: one rank of a world_size-2 gloo
group running the WHOLE configuration-recovery loop with ``seed=None``.  Only the control process may draw random
numbers and build CI strings (reference fermion.py:421-451); every rank must be handed the same batches and end
with the same result."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    import torch.distributed as dist

    from oracle import sqd_oracle as O
    from qiskit_addon_sqd_amd import _capi
    from qiskit_addon_sqd_amd.distributed import solve_sci_batch_distributed
    from qiskit_addon_sqd_amd.sqd import diagonalize_fermionic_hamiltonian

    _capi._LIB = _capi.bind(ctypes.CDLL(os.environ["SQD_EMU_LIB"]))
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}",
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank = dist.get_rank()
    norb, nelec = 6, (3, 3)
    h1, eri = O.synthetic_integrals(norb, seed=5)
    # noisy samples: every rank builds the same input matrix (an input, like the integrals), but with seed=None
    # the loop's own random stream differs from process to process -- unless only rank 0 uses it
    samples = np.random.default_rng(99).random((400, 2 * norb)) < 0.5
    seen = []

    def solver(ci_strings, one, two, norb_, nelec_):
        seen.append([(np.asarray(a).tolist(), np.asarray(b).tolist()) for a, b in ci_strings])
        return solve_sci_batch_distributed(ci_strings, one, two, norb_, nelec_, compute_rdms=False)

    calls = []
    res = diagonalize_fermionic_hamiltonian(h1, eri, samples, samples_per_batch=40, norb=norb, nelec=nelec,
                                            num_batches=3, max_iterations=3, sci_solver=solver, seed=None,
                                            callback=lambda r: calls.append(len(r)))
    assert (len(calls) > 0) == (rank == 0)   # the callback runs on the control process only (fermion.py:435)
    mine = (seen, float(res.energy), np.asarray(res.sci_state.amplitudes).tolist(),
            [np.asarray(o).tolist() for o in res.orbital_occupancies])
    everyone = [None, None]
    dist.all_gather_object(everyone, mine)
    assert everyone[0] == everyone[1], "ranks diverged"
    assert len(seen) >= 2 and all(len(batch) == 3 for batch in seen)
    # default solver in distributed mode = the collective one (no sci_solver argument)
    res2 = diagonalize_fermionic_hamiltonian(h1, eri, samples, samples_per_batch=40, norb=norb, nelec=nelec,
                                             num_batches=2, max_iterations=2, seed=None)
    e2 = [None, None]
    dist.all_gather_object(e2, float(res2.energy))
    assert e2[0] == e2[1]
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
