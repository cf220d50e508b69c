"""GPU tuning probe (not a test): per item class sigma time for large uniform sets.  argv: sizes"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
for n in [int(a) for a in sys.argv[1:]] or [8000, 10000]:
    sa, sb = S.uniform_strings(30, 8, n, 1001), S.uniform_strings(30, 8, n, 1001 + 7919)
    with _capi.Context(h1, eri) as ctx:
        ctx.set_subspace(sa, sb)
        row = []
        for mask in (0, 1, 2, 4, 7):
            os.environ['SQD_SIGMA_TYPES'] = str(mask)
            ctx.time_sigma(1)
            row.append(f"mask{mask}={ctx.time_sigma(4) * 1e3:8.1f}")
        os.environ.pop('SQD_SIGMA_TYPES')
        print('n', n, ' '.join(row), flush=True)
