"""GPU probe (not a test): aggregate throughput of 16 independent 317 x 317 batches on ONE GPU against the number of
solves in flight (``solve_sci_batch(concurrency=k)``), measured in the steady state: 0.3 s of spin-up per setting
(the process' first ~0.1 s of GPU activity contains one or two 30-50 ms stalls, profiles/r02/stall_probe.txt -- the
round-1 probe, one run per setting, caught them at random), then 7 repeats; median, min and max are reported."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd.fermion import solve_sci_batch

h1, eri = S.synthetic_integrals(30)
for name, gen in (('uniform', S.uniform_strings), ('hf', S.hf_centred_strings)):
    batches = [(gen(30, 8, 317, 100 + i), gen(30, 8, 317, 900 + i)) for i in range(16)]
    for k in (1, 2, 3, 4, 6, 8):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False, concurrency=k)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False, concurrency=k)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.array(ts)
        print(f'{name} 16 batches of 317x317, concurrency {k}: median {np.median(ts):.2f} ms (min {ts.min():.2f}, max {ts.max():.2f}) '
              f'= {np.median(ts) / 16:.3f} ms per batch', flush=True)
