"""GPU tuning probe (not a test): sigma time vs launch geometry knobs and per work-item type."""
import itertools, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
cases = (('hf', S.hf_centred_strings, 317), ('uniform', S.uniform_strings, 317), ('hf', S.hf_centred_strings, 1000), ('uniform', S.uniform_strings, 4000))
for name, gen, n in cases:
    sa, sb = gen(30, 8, n, 1001), gen(30, 8, n, 1001 + 7919)
    ctx = _capi.Context(h1, eri)
    for cap, K, T in itertools.product((8, 32), (2, 4, 8, 16), (0, 512, 1024)):
        os.environ['SQD_ELL_CAP'] = str(cap); os.environ['SQD_SIGMA_K'] = str(K)
        if T: os.environ['SQD_SIGMA_T'] = str(T)
        else: os.environ.pop('SQD_SIGMA_T', None)
        try:
            ctx.set_subspace(sa, sb)
            t = ctx.time_sigma(10) * 1e3
        except Exception as exc:
            t = float('nan')
        print(f"{name:8s} n={n:5d} cap={cap:3d} K={K:2d} T={T:4d}  sigma_us={t:8.1f}", flush=True)
    ctx.close()
