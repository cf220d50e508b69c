"""GPU probe (not a test): shifted solves of the device-side warm-started eigen-solver per Davidson iteration."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import synthetic as S, fermion as F
h1, eri = S.synthetic_integrals(30)
ctx = F._get_context(h1, eri, 0)
for name, gen, n in (('uniform', S.uniform_strings, 317), ('hf', S.hf_centred_strings, 317), ('hf', S.hf_centred_strings, 707)):
    sa, sb = gen(30, 8, n, 1000), gen(30, 8, n, 1000 + 7919)
    ctx.set_subspace(sa, sb)
    _, st = ctx.davidson(fetch=False)
    print(name, n, {k: st[k] for k in ('iterations', 'n_sigma', 'n_eig_solves', 'n_eig_fallbacks', 'converged')})
