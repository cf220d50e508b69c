"""GPU probe (not a test): the concurrency sweep of _concurrency_probe.py in a caller-given order (argv),
HF-centred batches only -- is the slow point a property of k or of the order the contexts were created in?"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd.fermion import solve_sci_batch
h1, eri = S.synthetic_integrals(30)
kind = os.environ.get('KIND', 'hf')
gen = S.hf_centred_strings if kind == 'hf' else S.uniform_strings
batches = [(gen(30, 8, 317, 100 + i), gen(30, 8, 317, 900 + i)) for i in range(16)]
for k in [int(a) for a in sys.argv[1:]]:
    solve_sci_batch(batches[:k], h1, eri, 30, (8, 8), compute_rdms=False, concurrency=k)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False, concurrency=k)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f'{kind} 16 batches, concurrency {k}: ' + ' '.join(f'{t:.1f}' for t in ts) + ' ms', flush=True)
