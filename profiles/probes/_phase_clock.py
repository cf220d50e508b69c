"""GPU probe (not a test): where a Davidson iteration's BLAS-1 kernels spend their time.  Needs the probe build of the
library (-DSQD_PHASE_CLOCK -> profiles/probes/_build/libsqd_hip_clk.so; the product library carries none of this):
the kernels add 100 MHz wall-clock deltas of their critical path into a device array, read here after N solves.
The marks wait for outstanding stores (s_waitcnt) where the product kernels do not: a phase reads up to ~1 us longer here."""
import ctypes as C, os, sys, time
from pathlib import Path
ROOT = Path(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
# (the clocks sit in k_dots_eig: the element-gather path is measured in its two-launch form)
os.environ.setdefault('SQD_DAV_FUSE_DIRECT', '0')
sys.path.insert(0, str(ROOT))
import numpy as np
from qiskit_addon_sqd_amd import _capi
_capi.LIB_PATH = ROOT / 'profiles' / 'probes' / '_build' / 'libsqd_hip_clk.so'
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import fermion as F

lib = _capi.load_library()
lib.sqd_probe_clk.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
buf = (C.c_ulonglong * (64 + 1024))()
for name, gen in (('hf', S.hf_centred_strings), ('uniform', S.uniform_strings)):
    sa, sb = gen(30, 8, 317, 1001), gen(30, 8, 317, 1001 + 7919)
    for _ in range(5):
        F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)
    lib.sqd_probe_clk(None, 1)
    t0 = time.perf_counter(); n = 20
    for _ in range(n):
        r = F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)
    wall = (time.perf_counter() - t0) / n * 1e3
    lib.sqd_probe_clk(buf, 1)
    c = np.array(buf[:], dtype=np.float64)
    us = lambda slot, cnt: c[slot] / max(cnt, 1) / 100.0
    nd, nr, no = c[0], c[10], c[20]
    print(f'{name} 317^2: {wall:.3f} ms per solve, {nd / n:.1f} k_dots_eig launches per solve, mean m {c[6] / max(nd, 1):.1f}')
    print(f'  k_dots_eig (the workgroup that arrives last): loop+sum+store {us(2, nd):.2f} us | arrival {us(3, nd):.2f} | '
          f'fold {us(4, nd):.2f} | eig step {us(5, nd):.2f} (state + matrix {us(7, nd):.2f}, eigenpair {us(8, nd):.2f})')
    print(f'  k_residual_precond (workgroup 0): state {us(11, nr):.2f} | loop {us(12, nr):.2f} | sum+store {us(13, nr):.2f}')
    if name == 'hf':
        b = c[64:64 + 2 * 197].reshape(-1, 2)
        t0 = b[:, 0].min()
        st, en = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0
        print('  iteration 10, k_dots_eig workgroups: start (us after the first) percentiles 0/25/50/75/100:', np.percentile(st, [0, 25, 50, 75, 100]).round(2),
              '| hand-over:', np.percentile(en, [0, 25, 50, 75, 100]).round(2), '| loop time:', np.percentile(en - st, [0, 50, 100]).round(2))
        print('   starts by workgroup index (every 8th):', st[::8].round(1))
    print(f'  k_orth_dev (workgroup 0): stop flag + fold {us(21, no):.2f} | decisions {us(22, no):.2f} | loop {us(23, no):.2f}', flush=True)
