"""Worker of tests/test_distributed.py::test_two_rank_gloo_row_sharded_sigma_and_solve: one rank of a world_size-2
gloo group holding HALF the alpha rows of one subspace.  The local sigma kernels run through the
kernel-logic emulator (device pointers are host pointers there); the all-gather / all-reduce logic is the real one."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    import torch
    import torch.distributed as dist

    from oracle import sqd_oracle as O
    from qiskit_addon_sqd_amd import _capi
    from qiskit_addon_sqd_amd.sharded import ShardedSubspace, row_range, solve_sci_sharded

    emu = _capi.bind(ctypes.CDLL(os.environ["SQD_EMU_LIB"]))
    _capi._LIB = emu
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}",
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    assert [row_range(13, r, 2) for r in range(2)] == [(0, 7), (7, 13)] and row_range(5, 3, 4) == (4, 5)

    norb, nelec = 7, (3, 3)
    h1, eri = O.synthetic_integrals(norb, seed=11)
    sa = O.hf_centred_strings(norb, 3, 13, 1)   # 13 rows over 2 ranks: ragged split 7 + 6
    sb = O.hf_centred_strings(norb, 3, 11, 2)
    H = O.build_php(h1, eri, sa, sb, norb)
    S2 = O.build_spin_square(sa, sb, norb, nelec)
    c = np.random.default_rng(3).standard_normal((13, 11))  # the same full vector on every rank

    sub = ShardedSubspace((sa, sb), h1, eri, lib=emu)
    lo, hi = sub.row0, sub.row1
    assert (lo, hi) == row_range(13, rank, world)
    shard = torch.from_numpy(c[lo:hi].copy())
    # (i) sharded sigma == rows of the single-context sigma == rows of the dense oracle
    with _capi.Context(h1, eri, lib=emu) as full:
        full.set_subspace(sa, sb)
        s_full = full.sigma(c)
        p_full = full.sigma(c, 1, 0.75, 0.3)
        ss_full = full.contract_ss(c)
        hd_full = full.hdiag()
    s_rows = sub.sigma(shard).numpy()
    assert np.array_equal(s_rows, s_full[lo:hi])                      # same kernels, same order: bit for bit
    assert np.allclose(s_rows.ravel(), (H @ c.ravel()).reshape(13, 11)[lo:hi].ravel(), atol=1e-12)
    assert np.array_equal(sub.sigma(shard, 1, 0.75, 0.3).numpy(), p_full[lo:hi])
    assert np.array_equal(sub.contract_ss(shard).numpy(), ss_full[lo:hi])
    assert np.array_equal(sub.hdiag.numpy(), hd_full[lo:hi])
    assert sub.n_allgather == 3
    # (ii) the gathered matrix is the full vector on every rank
    assert np.array_equal(sub.gather_rows(shard).numpy(), c)
    sub.close()

    # (iii) the collective solver against dense diagonalisation, without and with the spin penalty
    sz = 0.5 * abs(nelec[0] - nelec[1])
    for spin_sq in (None, 0.0, sz * (sz + 1.0) + 2.0):  # no penalty, pyscf's first form, the squared form (two gathers)
        res = solve_sci_sharded((sa, sb), h1, eri, norb, nelec, spin_sq=spin_sq, lib=emu)
        if spin_sq is None:
            Heff = H
        elif spin_sq < sz * (sz + 1.0) + 0.1:
            Heff = H + 0.2 * (S2 - spin_sq * np.eye(len(H)))
        else:
            P = S2 - spin_sq * np.eye(len(H))
            Heff = H + 0.2 * (P @ P)
        w, v = np.linalg.eigh(Heff)
        e_ref = float(v[:, 0] @ H @ v[:, 0])
        st = res._sharded_stats
        assert st["converged"] and st["n_allgather"] >= st["n_sigma"]
        assert abs(res.energy - e_ref) < 5e-7, (spin_sq, res.energy, e_ref)
        assert abs(abs(np.vdot(res.sci_state.amplitudes.ravel(), v[:, 0])) - 1.0) < 1e-6
        r1a, r1b = O.make_rdm1s(v[:, 0].reshape(13, 11), sa, sb, norb)
        assert np.allclose(res.orbital_occupancies[0], np.diag(r1a), atol=1e-5)
        assert np.allclose(res.orbital_occupancies[1], np.diag(r1b), atol=1e-5)
        # every rank holds the same answer
        both = [None, None]
        dist.all_gather_object(both, (float(res.energy), res.sci_state.amplitudes.tolist()))
        assert both[0] == both[1]
    # the two drivers -- the library's device-resident state machine (default) and the torch-level flow -- agree
    rn = solve_sci_sharded((sa, sb), h1, eri, norb, nelec, spin_sq=0.0, lib=emu, driver="native")
    rt = solve_sci_sharded((sa, sb), h1, eri, norb, nelec, spin_sq=0.0, lib=emu, driver="torch")
    assert rn._sharded_stats["converged"] and rt._sharded_stats["converged"]
    assert abs(rn.energy - rt.energy) < 5e-7  # (with a penalty <c|H|c> is first order in the residual, 1e-6)
    assert abs(abs(np.vdot(rn.sci_state.amplitudes, rt.sci_state.amplitudes)) - 1.0) < 1e-6
    # the sigma stage as two native calls around the all-gather (the default: the gather overlaps the own-row work items)
    # against the one-call stage behind the gather: the same bits, energy and state, with and without the linear penalty
    import qiskit_addon_sqd_amd.sharded as SH
    for spin_sq in (None, 0.0):
        outs = []
        for ov in ("1", "0"):
            os.environ["SQD_SHARD_OVERLAP"] = ov
            r = solve_sci_sharded((sa, sb), h1, eri, norb, nelec, spin_sq=spin_sq, lib=emu)
            outs.append((float(r.energy), r.sci_state.amplitudes.copy(), r._sharded_stats["n_sigma"]))
        os.environ.pop("SQD_SHARD_OVERLAP")
        assert outs[0][0] == outs[1][0] and outs[0][2] == outs[1][2] and np.array_equal(outs[0][1], outs[1][1]), spin_sq
    # sharded state only
    part = solve_sci_sharded((sa, sb), h1, eri, norb, nelec, gather_state=False, lib=emu)
    assert part.sci_state.amplitudes.shape == (hi - lo, 11)
    # whole-vector entry points refuse a sharded context
    with _capi.Context(h1, eri, lib=emu) as ctx:
        ctx.set_subspace_rows(sa, sb, 2, 9)
        try:
            ctx.davidson()
            raise AssertionError("davidson on a row shard must fail")
        except _capi.SQDNativeError as exc:
            assert "row shard" in str(exc)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
