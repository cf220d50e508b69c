"""GPU probe (not a test): where a work item of the sigma kernel spends its time.  Needs the probe build of the library
(-DSQD_PHASE_CLOCK -> profiles/probes/_build/libsqd_hip_clk.so): thread 0 of every workgroup adds the 100 MHz wall-clock
deltas of its item's phases into its own row of a device array.  Every mark waits for the workgroup's outstanding memory operations
(s_waitcnt 0), so overlap ACROSS phases that the product kernel has is not in these numbers: they say what each phase
costs when it runs alone in its workgroup, under the load of the other workgroups."""
import ctypes as C, os, sys, time
from pathlib import Path
ROOT = Path(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, str(ROOT))
import numpy as np
from qiskit_addon_sqd_amd import _capi
_capi.LIB_PATH = ROOT / 'profiles' / 'probes' / '_build' / 'libsqd_hip_clk.so'
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import fermion as F

lib = _capi.load_library()
lib.sqd_probe_clk_sigma.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
buf = (C.c_ulonglong * 64)()


def report(tag):
    lib.sqd_probe_clk_sigma(buf, 1)
    c = np.array(buf[:], dtype=np.float64)
    n0, n1 = max(c[0], 1), max(c[8], 1)
    us = lambda slot, n: c[slot] / n / 100.0
    print(f'{tag}: {int(c[0])} own-row items: stage {us(1, n0):.2f} us | virtual rows {us(2, n0):.2f} | sums + store {us(3, n0):.2f} | whole item {us(4, n0):.2f}')
    print(f'{" " * len(tag)}  {int(c[8])} alpha-single batch items: records {us(9, n1):.2f} us | rows staged {us(10, n1):.2f} | virtual rows (LDS gathers) {us(11, n1):.2f} | '
          f'sums + store {us(12, n1):.2f} | whole item {us(13, n1):.2f}', flush=True)


sa, sb = S.hf_centred_strings(30, 8, 317, 1001), S.hf_centred_strings(30, 8, 317, 1001 + 7919)
for _ in range(3):
    F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)
lib.sqd_probe_clk_sigma(None, 1)
for _ in range(5):
    F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)
report('single HF-centred 317^2')
batches = [(S.hf_centred_strings(30, 8, 317, 100 + i), S.hf_centred_strings(30, 8, 317, 900 + i)) for i in range(16)]
for _ in range(2):
    F.solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False)
lib.sqd_probe_clk_sigma(None, 1)
for _ in range(3):
    F.solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False)
report('16 x HF-centred 317^2, batched')
