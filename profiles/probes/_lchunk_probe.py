"""GPU tuning probe (not a test): sigma time vs links per AXPY item (L) and links folded into the own-row item (L0)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
h1b, erib = S.synthetic_integrals(40)
cases = (('hf', 30, 8, 317), ('hf', 30, 8, 1000), ('hf', 40, 15, 707), ('un', 30, 8, 4000))
for name, norb, ne, n in cases:
    gen = S.hf_centred_strings if name == 'hf' else S.uniform_strings
    sa, sb = gen(norb, ne, n, 1001), gen(norb, ne, n, 1001 + 7919)
    ctx = _capi.Context(*((h1, eri) if norb == 30 else (h1b, erib)))
    row = []
    for L0, L in ((16, 32), (16, 64), (16, 128), (16, 256), (32, 64), (32, 128), (64, 128), (0, 64)):
        os.environ['SQD_SIGMA_L0'] = str(L0); os.environ['SQD_SIGMA_LCHUNK'] = str(L)
        ctx.set_subspace(sa, sb)
        ctx.time_sigma(3)
        row.append(f"{L0}/{L}={ctx.time_sigma(20) * 1e3:7.1f}")
    print(name, norb, n, ' '.join(row), flush=True)
    ctx.close()
