"""GPU tuning probe (not a test): sigma time with / without the layout search."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
os.environ['SQD_DEBUG_GEOM'] = '1'
h1, eri = S.synthetic_integrals(30)
h1b, erib = S.synthetic_integrals(40)
cases = (('hf', 30, 8, 317), ('hf', 30, 8, 500), ('hf', 30, 8, 1000), ('hf', 30, 8, 2000), ('un', 30, 8, 317), ('hf', 40, 15, 707), ('un', 30, 8, 4000), ('un', 40, 15, 707))
for name, norb, ne, n in cases:
    gen = S.hf_centred_strings if name == 'hf' else S.uniform_strings
    sa, sb = gen(norb, ne, n, 1001), gen(norb, ne, n, 1001 + 7919)
    ctx = _capi.Context(*((h1, eri) if norb == 30 else (h1b, erib)))
    row = []
    for search in (0, 1):
        os.environ['SQD_SIGMA_SEARCH'] = str(search)
        ctx.set_subspace(sa, sb)
        ctx.time_sigma(3)
        row.append(f"search{search}={ctx.time_sigma(20) * 1e3:7.1f}")
    print(name, norb, n, ' '.join(row), flush=True)
    ctx.close()
