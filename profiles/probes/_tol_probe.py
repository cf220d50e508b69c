import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
h1b, erib = S.synthetic_integrals(40)
for name, norb, ne, n in (('hf', 30, 8, 317), ('un', 30, 8, 317), ('hf', 40, 15, 707), ('hf', 30, 8, 1000)):
    gen = S.hf_centred_strings if name == 'hf' else S.uniform_strings
    sa, sb = gen(norb, ne, n, 1001), gen(norb, ne, n, 1001 + 7919)
    with _capi.Context(*((h1, eri) if norb == 30 else (h1b, erib))) as ctx:
        ctx.set_subspace(sa, sb)
        _, st0 = ctx.davidson(tol_residual=1e-9, max_cycle=300, fetch=False); e_ref = ctx.energy()
        for tr in (None, 3.1622776601683795e-05):
            _, st = ctx.davidson(tol_residual=tr, fetch=False)
            e = ctx.energy()
            print(name, norb, n, 'tol_residual', tr, 'n_sigma', st['n_sigma'], 'resid', f"{st['residual']:.2e}", 'E-Eref', f"{e - e_ref:.2e}", 'ms', round(st['ms_total'], 3), flush=True)
