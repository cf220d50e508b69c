"""GPU probe (not a test): where a workgroup of k_opp_src spends its time.  Needs the probe build of the library
(profiles/probes/build_variant.sh oclk "-DSQD_PHASE_CLOCK" sqd_oppsrc.hip -> profiles/probes/_build/libsqd_hip_clk.so).
env N (strings per spin), plus the kernel's hooks (SQD_OPP_SRC=1 forces it below 3073 columns)."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, str(ROOT))
import numpy as np
from qiskit_addon_sqd_amd import _capi
_capi.LIB_PATH = ROOT / "profiles" / "probes" / "_build" / "libsqd_hip_clk.so"
from qiskit_addon_sqd_amd import synthetic as S

lib = _capi.load_library()
lib.sqd_probe_clk_oppsrc.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n = int(os.environ.get('N', '3000'))
h1, eri = S.synthetic_integrals(30)
sa, sb = S.hf_centred_strings(30, 8, n, 11), S.hf_centred_strings(30, 8, n, 13)
buf = (C.c_ulonglong * 8)()
with _capi.Context(h1, eri) as ctx:
    ctx.set_subspace(sa, sb)
    ctx.sync()
    print(ctx.sigma_kernel(), flush=True)
    ctx.time_sigma(2)
    lib.sqd_probe_clk_oppsrc(None, 1)
    reps = 5
    t = ctx.time_sigma(reps)
    lib.sqd_probe_clk_oppsrc(buf, 0)
    c = np.array(buf[:], dtype=np.float64)
    passes, items, rounds = c[0] / reps, c[6] / reps, c[7] / reps
    us = lambda k: c[k] / 100.0 / reps
    print(f"hf {n}^2: sigma {1e3 * t:.1f} us; per sigma: {items:.0f} items, {passes:.0f} passes, {rounds:.0f} rounds")
    print(f"  summed over workgroups (us): stage {us(1):.0f} | gather {us(2):.0f} | fold: scatter {us(3):.0f} + column sums {us(4):.0f} | whole items {us(5):.0f}")
    print(f"  per round: stage {us(1) / rounds:.2f} us, gather {us(2) / rounds:.2f} us; per pass: fold {(us(3) + us(4)) / passes:.2f} us "
          f"({us(3) / passes:.2f} + {us(4) / passes:.2f}); per item {us(5) / items:.2f} us; workgroup slots busy: {us(5) / (1e3 * t):.0f}")
