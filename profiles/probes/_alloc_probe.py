import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd.fermion import solve_sci_batch
h1, eri = S.synthetic_integrals(30)
for name, gen in (('uniform', S.uniform_strings), ('hf', S.hf_centred_strings)):
    batches = [(gen(30, 8, 317, 100 + i), gen(30, 8, 317, 900 + i)) for i in range(16)]
    for k in (1, 2, 4, 6, 8):
        print(f'--- {name} k={k} warm-up', file=sys.stderr, flush=True)
        solve_sci_batch(batches[:k], h1, eri, 30, (8, 8), compute_rdms=False, concurrency=k)
        print(f'--- {name} k={k} timed', file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False, concurrency=k)
        dt = time.perf_counter() - t0
        print(f'{name} k={k}: {dt*1e3:.2f} ms', file=sys.stderr, flush=True)
