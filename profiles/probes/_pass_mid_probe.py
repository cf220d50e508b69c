"""GPU probe (not a test): mid-size rows (4000..8000 strings) with the default layout vs the multi-pass walk
forced (SQD_SIGMA_PASS in the environment).  argv: kind sizes..."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
kind = sys.argv[1]
gen = S.hf_centred_strings if kind == 'hf' else S.uniform_strings
for n in [int(a) for a in sys.argv[2:]]:
    sa, sb = gen(30, 8, n, 1001), gen(30, 8, n, 1001 + 7919)
    with _capi.Context(h1, eri) as ctx:
        ctx.set_subspace(sa, sb); ctx.hdiag()
        t = ctx.time_sigma(3) * 1e3
        print(f"{kind:8s} n={n:6d} pass={os.environ.get('SQD_SIGMA_PASS','-'):>5s} sigma_us={t:10.1f}", flush=True)
