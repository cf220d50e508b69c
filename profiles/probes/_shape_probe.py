"""GPU probe: HF-centred na x nb sets of unequal sides -- which sigma formulation the selection takes and what it costs
beside the forced alternatives.  env SHAPES="300x3000 ...", MODES."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qiskit_addon_sqd_amd import _capi, synthetic as S
shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "300x3000 3000x300 600x6000 6000x600 900x8000 8000x900 2000x10000 10000x2000").split()]
modes = {"default": {}, "items": {"SQD_SIGMA_SPMM": "0", "SQD_SIGMA_DENSE": "0"}, "mfma": {"SQD_SIGMA_SPMM": "0", "SQD_SIGMA_DENSE": "1"},
         "spmm": {"SQD_SIGMA_SPMM": "1"}}
h1, eri = S.synthetic_integrals(30)
for na, nb in shapes:
    sa, sb = S.hf_centred_strings(30, 8, na, 11), S.hf_centred_strings(30, 8, nb, 13)
    print('strings ready', na, nb, flush=True)
    out = []
    for name in os.environ.get("MODES", "default items mfma spmm").split():
        for k in ("SQD_SIGMA_SPMM", "SQD_SIGMA_DENSE", "SQD_SIGMA_OPP"):
            os.environ.pop(k, None)
        os.environ.update(modes[name])
        try:
            with _capi.Context(h1, eri) as ctx:
                ctx.set_subspace(sa, sb)
                ctx.time_sigma(1)
                out.append(f"{name}: {ctx.sigma_kernel()} {1e3 * ctx.time_sigma(3):9.1f} us")
                print("   ", na, nb, out[-1], flush=True)
        except Exception as exc:  # noqa: BLE001
            out.append(f"{name}: FAILED {exc!r}"[:90])
    print(f"{na:6d} x {nb:6d} D={na * nb:.1e}  " + " | ".join(out), flush=True)
