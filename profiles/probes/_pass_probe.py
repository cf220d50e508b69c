"""GPU probe (not a test): sigma time where the multi-pass walk of the beta lists applies (row fits LDS,
one partial sum per virtual row does not), against the global-row path (SQD_SIGMA_NOPASS=1), plus the
tuned sizes as a regression check.  argv: sizes..."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
sizes = [int(a) for a in sys.argv[1:]] or [317, 1000, 4000, 8000, 10000, 12000]
for name, gen in (('uniform', S.uniform_strings), ('hf', S.hf_centred_strings)):
    for n in sizes:
        if name == 'hf' and n > 4000:
            continue
        sa, sb = gen(30, 8, n, 1001), gen(30, 8, n, 1001 + 7919)
        with _capi.Context(h1, eri) as ctx:
            t0 = time.time(); ctx.set_subspace(sa, sb); ctx.hdiag(); t1 = time.time()
            t = ctx.time_sigma(5 if n >= 4000 else 20) * 1e3
            print(f"{name:8s} n={n:6d} D={n*n:.2e} setup_s={t1-t0:6.2f} sigma_us={t:10.1f} "
                  f"alg_GBs={ctx.sigma_bytes()/t/1e3:8.1f}", flush=True)
