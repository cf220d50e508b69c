"""GPU probe (not a test): where a workgroup of the list-pass sigma kernel (sqd_lists.hip) spends its row loop.  Needs the
probe build of the library (-DSQD_PHASE_CLOCK -> profiles/probes/_build/libsqd_hip_clk.so): thread 0 of every workgroup adds
the 100 MHz wall-clock deltas of the loop's phases into its own row of a device array."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, str(ROOT))
import numpy as np
from qiskit_addon_sqd_amd import _capi
_capi.LIB_PATH = Path(os.environ.get('SQD_LIB', str(ROOT / 'profiles' / 'probes' / '_build' / 'libsqd_hip_clk.so')))
from qiskit_addon_sqd_amd import synthetic as S, fermion as F

lib = _capi.load_library()
lib.sqd_probe_clk_lists.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n = int(os.environ.get('N', '10000'))
h1, eri = S.synthetic_integrals(30)
ctx = F._get_context(h1, eri, 0)
sa, sb = S.uniform_strings(30, 8, n, 11), S.uniform_strings(30, 8, n, 13)
ctx.set_subspace(sa, sb)
print(ctx.sigma_kernel(), 'sigma', ctx.time_sigma(3), 'ms')
buf = (C.c_ulonglong * 32)()
lib.sqd_probe_clk_lists(None, 1)
reps = 4
ms = ctx.time_sigma(reps - 1)  # (time_sigma runs one warm-up application: reps applications in all)
lib.sqd_probe_clk_lists(buf, 0)
c = np.array(buf[:], dtype=np.float64).reshape(4, 8)
names = ['requests', 'row from LDS', 'barrier 1', 'next row -> LDS', 'epilogue', 'barrier 2']
nwg = 240.0  # (8 XCDs x 3 row chunks x 10 column blocks at 10^4 x 10^4)
for v, tag in ((1, 'list pass (beta side, H)'),):
    tot = c[v, :6].sum()
    print(f'{tag}: per workgroup and launch {tot / nwg / reps / 100.0:.1f} us  |  ' +
          '  '.join(f'{nm} {100.0 * c[v, i] / max(tot, 1):.1f}%' for i, nm in enumerate(names)), flush=True)
