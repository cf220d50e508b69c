"""GPU probe (not a test): one sigma at uniform (GEN=hf: HF-centred) N x N under the tuning hooks given in the environment."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import synthetic as S, fermion as F, _capi
if os.environ.get('SQD_LIB'):  # another build of the library (tuning probes)
    from pathlib import Path
    _capi.LIB_PATH = Path(os.environ['SQD_LIB'])
n = int(os.environ.get('N', '10000'))
h1, eri = S.synthetic_integrals(30)
ctx = F._get_context(h1, eri, 0)
gen = S.hf_centred_strings if os.environ.get('GEN') == 'hf' else S.uniform_strings
sa, sb = gen(30, 8, n, 11), gen(30, 8, n, 13)
t0 = time.perf_counter(); ctx.set_subspace(sa, sb); ctx.sync(); t1 = time.perf_counter()
ms = ctx.time_sigma(5)
b = ctx.sigma_bytes()
print({k: os.environ[k] for k in os.environ if k.startswith('SQD_')}, ctx.sigma_kernel(), f'n={n} set_subspace {1e3*(t1-t0):.1f} ms  sigma {ms:.3f} ms  '
      f'{b/ms/1e6:.0f} GB/s  frac {b/ms/1e6/8000:.4f}', flush=True)
