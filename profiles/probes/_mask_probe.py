"""GPU tuning probe (not a test): sigma time per work-item class (SQD_SIGMA_TYPES mask) -- k_sigma only and with reduce."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
for name, gen, n in (('hf', S.hf_centred_strings, 317), ('un', S.uniform_strings, 317), ('hf', S.hf_centred_strings, 1000)):
    sa, sb = gen(30, 8, n, 1001), gen(30, 8, n, 1001 + 7919)
    with _capi.Context(h1, eri) as ctx:
        ctx.set_subspace(sa, sb)
        row = []
        for mask in (0, 1, 2, 4, 3, 7):
            os.environ['SQD_SIGMA_TYPES'] = str(mask)
            ctx.time_sigma(3)
            row.append(f"mask{mask}={ctx.time_sigma(20) * 1e3:7.1f}")
        os.environ.pop('SQD_SIGMA_TYPES')
        print(name, n, ' '.join(row), flush=True)
