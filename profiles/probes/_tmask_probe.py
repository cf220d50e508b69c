"""GPU tuning probe (not a test): per item class sigma time at T = 256 / 512 / 1024 threads (hf 317)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
sa, sb = S.hf_centred_strings(30, 8, 317, 1001), S.hf_centred_strings(30, 8, 317, 1001 + 7919)
with _capi.Context(h1, eri) as ctx:
    for T in (320, 512, 768, 1024):
        os.environ['SQD_SIGMA_T'] = str(T)
        ctx.set_subspace(sa, sb)
        row = []
        for mask in (0, 1, 2, 4, 7):
            os.environ['SQD_SIGMA_TYPES'] = str(mask)
            ctx.time_sigma(3)
            row.append(f"mask{mask}={ctx.time_sigma(20) * 1e3:6.1f}")
        os.environ.pop('SQD_SIGMA_TYPES')
        print('T', T, ' '.join(row), flush=True)
