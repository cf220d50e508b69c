"""GPU-box probe (host side): the native digest of the integral tensors (sqd_hash_start / sqd_hash_finish) by number of
hash threads, against a Python-side xxh3 pass; run once per setting (the pool reads SQD_HASH_THREADS when it starts)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import fermion as F
rng = np.random.default_rng(0)
e = rng.standard_normal((30, 30, 30, 30)); h = rng.standard_normal((30, 30))
F._native_digests(e.reshape(-1), h.reshape(-1))
ts = []
for _ in range(300):
    t = time.perf_counter(); F._native_digests(e.reshape(-1), h.reshape(-1)); ts.append(time.perf_counter() - t)
    time.sleep(0.0001)  # (a solve's worth of pause between jobs)
ts.sort()
line = f"SQD_HASH_THREADS={os.environ.get('SQD_HASH_THREADS', 'default')}: native digest of 6.5 MB + 7 KB  median {1e3*ts[150]:.3f} ms  p90 {1e3*ts[270]:.3f} ms"
try:
    import xxhash
    t = time.perf_counter()
    for _ in range(100): xxhash.xxh3_64_intdigest(e.reshape(-1).data)
    line += f"  | xxh3 (Python, one core, GIL held) {1e3*(time.perf_counter()-t)/100:.3f} ms"
except ImportError:
    pass
print(line, f"| logical CPUs {os.cpu_count()}")
