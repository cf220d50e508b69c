"""GPU probe (not a test): sigma time on the global-row path (rows too long for LDS)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
for name, gen in (('uniform', S.uniform_strings), ('hf', S.hf_centred_strings)):
    for na, nb in ((317, 20000), (2000, 20000), (20000, 317)):
        sa, sb = gen(30, 8, na, 1001), gen(30, 8, nb, 1001 + 7919)
        with _capi.Context(h1, eri) as ctx:
            t0 = time.time(); ctx.set_subspace(sa, sb); ctx.hdiag(); t1 = time.time()
            t = ctx.time_sigma(5) * 1e3
            print(f"{name:8s} na={na:6d} nb={nb:6d} D={na*nb:.2e} setup_s={t1-t0:6.2f} sigma_us={t:10.1f} "
                  f"alg_GBs={ctx.sigma_bytes()/t/1e3:8.1f}", flush=True)
