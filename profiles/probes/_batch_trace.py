"""GPU probe for rocprofv3 --kernel-trace: a handful of batched solves of one configuration.
env CASE = uniform8 | uniform16 | hf8 | hf16 ; MODE = batched | serial"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import fermion as F

case = os.environ.get('CASE', 'hf16')
gen = S.hf_centred_strings if case.startswith('hf') else S.uniform_strings
nb = int(case.lstrip('hfuniorm'))
h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
batches = [(gen(30, 8, 317, 100 + i), gen(30, 8, 317, 900 + i)) for i in range(nb)]
kw = dict(concurrency=1) if os.environ.get('MODE', 'batched') == 'serial' else {}
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    F.solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False, **kw)
for _ in range(int(os.environ.get('REPS', '5'))):
    F.solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False, **kw)
