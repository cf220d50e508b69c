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
    # A solve that fails on ONE rank raises on EVERY rank, behind the exchange -- nobody is left waiting in the collective
    # or in the state transfer that follows: the owner raises its own exception, the others the RuntimeError that names
    # the failed batches.  Cases: the bad batch on rank 1, on the control process (rank 0), a batch whose strings are
    # consistent among themselves (the native build accepts them) but carry the wrong electron number in either
    # direction (4 electrons: it WOULD have been the lowest energy) -- caught on the host before the solve, i.e. before
    # the solve's hook could enqueue the exchange.
    def expect_failure(bad_batches, bad_index, owner_text):
        owner = bad_index % world
        try:
            solve_sci_batch_distributed(bad_batches, h1, eri, norb, nelec, compute_rdms=False)
        except ValueError as exc:
            assert rank == owner, (rank, exc)
            assert owner_text in str(exc), str(exc)
        except RuntimeError as exc:
            assert rank != owner, (rank, exc)
            assert "failed on ranks" in str(exc) and f"[{bad_index}]" in str(exc) and f"[{owner}]" in str(exc), str(exc)
        else:
            raise AssertionError(f"rank {rank}: expected the failing batch {bad_index} to raise on every rank")
        again = solve_sci_batch_distributed(batches, h1, eri, norb, nelec, compute_rdms=False)  # the group still works
        assert all(a.energy == s.energy for a, s in zip(again, serial))

    mixed = (np.array([1, 2, 4]), np.array([1, 2, 4]))  # Hamming weight 1, nelec says 3
    four = (O.random_strings(norb, 4, 6, 31), O.random_strings(norb, 4, 5, 32))
    two = (O.random_strings(norb, 2, 6, 33), O.random_strings(norb, 2, 5, 34))
    expect_failure([batches[0], mixed, batches[2]], 1, "Hamming")
    expect_failure([mixed, batches[1], batches[2]], 0, "Hamming")
    expect_failure([batches[0], four, batches[2]], 1, "Hamming")
    expect_failure([two, batches[1], batches[2]], 0, "Hamming")
    # ... and a failure the owner cannot announce: its kernels leave a record without a state (c.c = 0) and the call
    # returns -- on a GPU the owner's "zero norm" error comes after its hook has enqueued the exchange.  Simulated by
    # wiping the record behind rank 1's solve: every rank, the owner included, reads the same reduced table and raises.
    if rank == 1:
        ptrs = {}
        orig_set, orig_solve = _capi.Context.set_record_out, _capi.Context.solve

        def set_rec(self, ptr, stride=0):
            if ptr:
                ptrs["p"] = ptr
            return orig_set(self, ptr, stride)

        def solve_then_wipe(self, *a, **k):
            out = orig_solve(self, *a, **k)
            ctypes.memset(ptrs["p"], 0, 8 * _capi.record_width(norb))
            return out

        _capi.Context.set_record_out, _capi.Context.solve = set_rec, solve_then_wipe
    try:
        solve_sci_batch_distributed(batches[:2], h1, eri, norb, nelec, compute_rdms=False)
        raise AssertionError("expected the state-less record to raise on every rank")
    except RuntimeError as exc:
        assert "failed on ranks" in str(exc) and "[1]" in str(exc), str(exc)
    if rank == 1:
        _capi.Context.set_record_out, _capi.Context.solve = orig_set, orig_solve
    again = solve_sci_batch_distributed(batches, h1, eri, norb, nelec, compute_rdms=False)
    assert all(a.energy == s.energy for a, s in zip(again, serial))
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
