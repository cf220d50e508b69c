"""Worker of tests/test_distributed.py::test_one_rank_group_single_call_iteration: a gloo group of ONE rank.  The
row-sharded solver then holds all rows and runs every Davidson iteration as ONE native call
(``sqd_shard_dav_iteration``: pick + sigma + fused dots / eigen kernel + residual + orth); with
SQD_SHARD_FORCE_COLLECTIVES=2 it runs the staged calls of a real group around the (identity) collectives.  Same bits,
and the single-context solver's energy."""
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
    from qiskit_addon_sqd_amd.sharded import solve_sci_sharded

    emu = _capi.bind(ctypes.CDLL(os.environ["SQD_EMU_LIB"]))
    _capi._LIB = emu
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=0, world_size=1)
    norb, nelec = 7, (3, 3)
    h1, eri = O.synthetic_integrals(norb, seed=11)
    sa, sb = O.hf_centred_strings(norb, 3, 13, 1), O.hf_centred_strings(norb, 3, 11, 2)
    H = O.build_php(h1, eri, sa, sb, norb)
    e0 = np.linalg.eigvalsh(H)[0]
    out = {}
    for spin_sq in (None, 0.0):
        for force in ("", "2"):
            if force:
                os.environ["SQD_SHARD_FORCE_COLLECTIVES"] = force
            else:
                os.environ.pop("SQD_SHARD_FORCE_COLLECTIVES", None)
            out[spin_sq, force] = solve_sci_sharded((sa, sb), h1, eri, norb, nelec, spin_sq=spin_sq, lib=emu)
        a, b = out[spin_sq, ""], out[spin_sq, "2"]
        assert a._sharded_stats["converged"] and b._sharded_stats["converged"]
        assert a._sharded_stats["n_sigma"] == b._sharded_stats["n_sigma"]
        assert a.energy == b.energy and np.array_equal(a.sci_state.amplitudes, b.sci_state.amplitudes), (spin_sq, a.energy, b.energy)
    assert abs(out[None, ""].energy - e0) < 1e-8, (out[None, ""].energy, e0)
    dist.destroy_process_group()
    print("rank 0 ok")


if __name__ == "__main__":
    main()
