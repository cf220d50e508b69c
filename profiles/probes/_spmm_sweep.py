"""GPU probe (not a test): the sparse-product same-spin path (sqd_spmm.hip) against its tuning hooks.
One process per setting (the hooks are read once per process): env SIZES, then J / XCD / ORDER from the command line."""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    from qiskit_addon_sqd_amd import _capi, synthetic as S

    n = int(sys.argv[2])
    h1, eri = S.synthetic_integrals(30)
    sa, sb = S.hf_centred_strings(30, 8, n, 11), S.hf_centred_strings(30, 8, n, 13)
    with _capi.Context(h1, eri) as ctx:
        ctx.set_subspace(sa, sb)
        ctx.time_sigma(2)
        print(f"n={n} J={os.environ.get('SQD_SPMM_J')} xcd={os.environ.get('SQD_SPMM_XCD')} order={os.environ.get('SQD_SPMM_ORDER')} "
              f"types={os.environ.get('SQD_SIGMA_TYPES')} kernel={ctx.sigma_kernel()} sigma_us={1e3 * ctx.time_sigma(5):9.1f}", flush=True)
    sys.exit(0)
for n in os.environ.get("SIZES", "1000 3000").split():
    for J in ("1", "2", "4"):
        for xcd in ("0", "1"):
            for order in ("0", "1"):
                env = dict(os.environ, SQD_SIGMA_SPMM="1", SQD_SPMM_J=J, SQD_SPMM_XCD=xcd, SQD_SPMM_ORDER=order, SQD_SIGMA_TYPES="0")
                subprocess.run([sys.executable, __file__, "one", n], env=env)
