"""GPU probe (not a test): CONNECTED (HF-centred) subspaces at D = 1e6 .. 1e7 -- what the sigma kernels cost there.

env SIZES="1000 2000 3000" (strings per spin), MODES="default dense0 dense1", REPS, CHECK=1 (sampled rows against the
row-restricted string-space oracle), DAV=1 (one whole Davidson run).  One line per (size, mode)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qiskit_addon_sqd_amd import _capi, synthetic as S  # noqa: E402

if os.environ.get("SQD_LIB"):  # A/B against another build of the library (profiles/probes/build_variant.sh)
    from pathlib import Path

    _capi.LIB_PATH = Path(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")) / os.environ["SQD_LIB"]

sizes = [int(s) for s in os.environ.get("SIZES", "1000 2000 3000").split()]
modes = os.environ.get("MODES", "default dense0 dense1 spmm0").split()
reps = int(os.environ.get("REPS", "5"))
check = os.environ.get("CHECK", "1") == "1"
dav = os.environ.get("DAV", "1") == "1"
norb, nel = int(os.environ.get("NORB", "30")), int(os.environ.get("NELEC", "8"))
h1, eri = S.synthetic_integrals(norb)
for n in sizes:
    sa, sb = S.hf_centred_strings(norb, nel, n, 11), S.hf_centred_strings(norb, nel, n, 13)
    x = np.random.default_rng(3).standard_normal((n, n))
    ref = rows = None
    for mode in modes:
        os.environ.pop("SQD_SIGMA_DENSE", None)
        os.environ.pop("SQD_SIGMA_CONN", None)
        os.environ.pop("SQD_SIGMA_SPMM", None)
        if mode == "dense0":
            os.environ["SQD_SIGMA_DENSE"] = "0"
        elif mode == "dense1":
            os.environ["SQD_SIGMA_DENSE"] = "1"
        elif mode == "spmm0":
            os.environ["SQD_SIGMA_SPMM"] = "0"
        elif mode == "spmm1":
            os.environ["SQD_SIGMA_SPMM"] = "1"
        elif mode == "conn0":
            os.environ["SQD_SIGMA_CONN"] = "0"
        elif mode == "conn1":
            os.environ["SQD_SIGMA_CONN"] = "1"
        with _capi.Context(h1, eri) as ctx:
            try:
                ctx.set_subspace(sa, sb)
                ctx.sync()
                t0 = time.perf_counter()
                ctx.set_subspace(sa, sb)
                ctx.sync()
                t_tab = 1e3 * (time.perf_counter() - t0)
                kern = ctx.sigma_kernel()
                ctx.time_sigma(2)
                t_sig = ctx.time_sigma(reps)
                b = ctx.sigma_bytes()
                line = (f"hf n={n} mode={mode:8s} kernel={kern:22s} tables_ms={t_tab:8.2f} sigma_us={1e3 * t_sig:9.1f} "
                        f"B_sigma_MB={b / 1e6:8.1f} GBs={b / (t_sig * 1e-3) / 1e9:8.1f} frac={b / (t_sig * 1e-3) / 1e9 / 8000:6.4f} "
                        f"links={ctx.link_counts(0)},{ctx.link_counts(1)}")
                if dav:
                    ctx.davidson(fetch=False)
                    ctx.sync()
                    t0 = time.perf_counter()
                    _, st = ctx.davidson(fetch=False)
                    ctx.sync()
                    ms = 1e3 * (time.perf_counter() - t0)
                    line += (f" davidson_ms={ms:8.2f} n_sigma={st['n_sigma']} us_per_iter={1e3 * ms / max(st['n_sigma'], 1):8.1f} "
                             f"conv={st['converged']} e={st['e_davidson']:.10f}")
                if check:
                    from oracle import sqd_oracle as O

                    if ref is None:
                        rng = np.random.default_rng(5)
                        rows = np.unique(np.concatenate(([0, 1, n - 1], rng.choice(n, 9, replace=False))))
                        ref = O.sigma_rows_string_space(h1, eri, sa, sb, x, norb, rows)
                    sx = ctx.sigma(x)
                    err = np.abs(sx[rows] - ref).max() / max(1.0, np.abs(ref).max())
                    line += f" rows_rel_err={err:.2e} repro={bool(np.array_equal(sx, ctx.sigma(x)))}"
                print(line, flush=True)
            except Exception as exc:  # keep going: the probe is a survey
                print(f"hf n={n} mode={mode} FAILED: {exc!r}", flush=True)
