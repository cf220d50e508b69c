"""GPU probe (not a test): the batched native solve (sqd_solve_batch through solve_sci_batch) against the one-by-one and
the threads-x-streams paths: 8 uniform / 16 HF-centred 317 x 317 subspaces of the N2-sized problem on ONE GPU."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import fermion as F

if os.environ.get("SQD_LIB"):  # A/B against another build of the library (profiles/probes/build_variant.sh)
    from pathlib import Path
    from qiskit_addon_sqd_amd import _capi
    _capi.LIB_PATH = Path(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")) / os.environ["SQD_LIB"]

h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
for name, gen, nb in (('uniform', S.uniform_strings, 8), ('uniform', S.uniform_strings, 16), ('hf', S.hf_centred_strings, 8),
                      ('hf', S.hf_centred_strings, 16)):
    batches = [(gen(30, 8, 317, 100 + i), gen(30, 8, 317, 900 + i)) for i in range(nb)]
    for mode, kw in (('one by one', dict(concurrency=1)), ('4 streams', dict(concurrency=4)), ('batched', dict())):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            F.solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False, **kw)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            F.solve_sci_batch(batches, h1, eri, 30, (8, 8), compute_rdms=False, **kw)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.array(ts)
        nsig = sum(s['n_sigma'] for s in F._TLS.batch_stats) if mode == 'batched' else 0
        print(f'{name} {nb} x 317^2, {mode:10s}: median {np.median(ts):7.3f} ms (min {ts.min():.3f}, max {ts.max():.3f}) = '
              f'{np.median(ts) / nb:.4f} ms per batch' + (f'  [{nsig} sigma builds]' if nsig else ''), flush=True)
