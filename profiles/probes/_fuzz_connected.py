"""GPU probe (not a test): the default sigma formulation on CONNECTED subspaces of odd shapes -- unequal string counts on
either side of the selection thresholds, unequal electron numbers, few / many orbitals (up to 64), mixed HF-centred +
uniform sets -- against (a) the work-item formulation of the same library and (b) the row-restricted string-space oracle
on sampled rows, plus bit-reproducibility and one Davidson run per case (residual / Rayleigh check).  env CASES, SEED."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import sqd_oracle as O  # noqa: E402
from qiskit_addon_sqd_amd import _capi, synthetic as S  # noqa: E402

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ncases = int(os.environ.get("CASES", "14"))
MODE_ENV = ("SQD_SIGMA_DENSE", "SQD_SIGMA_SPMM", "SQD_SIGMA_OPP", "SQD_SIGMA_CONN")


def strings(norb, ne, n, kind, seed):
    cap = 1
    for k in range(ne):
        cap = cap * (norb - k) // (k + 1)
    n = min(n, cap)
    if kind == "hf":
        return S.hf_centred_strings(norb, ne, n, seed)
    hf = S.hf_centred_strings(norb, ne, max(2, n // 2), seed)
    uni = S.uniform_strings(norb, ne, n, seed + 1) if hasattr(S, "uniform_strings") else hf
    return np.unique(np.concatenate([hf, uni]))[:n] if len(uni) else hf


def sigma_with(env, h1, eri, sa, sb, x):
    for k in MODE_ENV:
        os.environ.pop(k, None)
    os.environ.update(env)
    with _capi.Context(h1, eri) as ctx:
        ctx.set_subspace(sa, sb)
        kern = ctx.sigma_kernel()
        y = ctx.sigma(x)
        y2 = ctx.sigma(x)
        t = ctx.time_sigma(3)
        pen = [ctx.sigma(x, use_spin=m, ss=0.75, shift=0.3) for m in (1, 2)] + [ctx.contract_ss(x)]
        return kern, y, bool(np.array_equal(y, y2)), t, pen


bad = 0
for case in range(ncases):
    norb = int(rng.choice([14, 18, 24, 30, 36, 48, 60]))
    nea = int(rng.integers(3, min(10, norb // 2) + 1))
    neb = nea if rng.random() < 0.5 else int(rng.integers(2, min(10, norb // 2) + 1))
    na = int(rng.choice([300, 880, 896, 900, 1023, 1024, 1025, 1400, 2100]))
    nb = na if rng.random() < 0.4 else int(rng.choice([200, 890, 896, 1000, 1024, 1300, 1900]))
    if os.environ.get("NA_LIST"):  # unequal sides: env NA_LIST / NB_LIST
        na = int(rng.choice([int(v) for v in os.environ["NA_LIST"].split()]))
        nb = int(rng.choice([int(v) for v in os.environ["NB_LIST"].split()]))
    kind = "hf" if rng.random() < 0.7 else "mixed"
    h1, eri = S.synthetic_integrals(norb)
    sa, sb = strings(norb, nea, na, kind, 100 + case), strings(norb, neb, nb, kind, 200 + case)
    na, nb = len(sa), len(sb)
    x = np.random.default_rng(case).standard_normal((na, nb))
    t0 = time.perf_counter()
    try:
        kern, y, repro, t_def, pen = sigma_with({}, h1, eri, sa, sb, x)
        kern_i, y_i, repro_i, t_it, pen_i = sigma_with({"SQD_SIGMA_SPMM": "0", "SQD_SIGMA_DENSE": "0"}, h1, eri, sa, sb, x)
        e_pen = max(np.abs(a - b).max() / max(1.0, np.abs(b).max()) for a, b in zip(pen, pen_i))
        rows = np.unique(np.concatenate(([0, na - 1], rng.choice(na, min(4, na), replace=False))))
        ref = O.sigma_rows_string_space(h1, eri, sa, sb, x, norb, rows)
        scale = max(1.0, np.abs(ref).max())
        e_or = np.abs(y[rows] - ref).max() / scale
        e_it = np.abs(y - y_i).max() / max(1.0, np.abs(y_i).max())
        for k in MODE_ENV:
            os.environ.pop(k, None)
        with _capi.Context(h1, eri) as ctx:
            ctx.set_subspace(sa, sb)
            c, st = ctx.davidson()
            hc = ctx.sigma(c)
            e = float(np.vdot(c, hc))
            res = float(np.linalg.norm(hc - e * c))
        ok = e_or < 1e-12 and e_it < 1e-12 and e_pen < 1e-12 and repro and repro_i and st["converged"] and res < 5e-4 and abs(e - st["e_davidson"]) < 1e-8
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} case {case:2d} norb={norb:2d} nelec=({nea},{neb}) {kind:5s} {na:5d} x {nb:5d} {kern:24s} "
              f"{1e3 * t_def:8.1f} us (items {kern_i}: {1e3 * t_it:8.1f} us)  vs oracle rows {e_or:.1e}  vs items {e_it:.1e}  penalty forms + S^2 vs items {e_pen:.1e}  "
              f"repro {repro}/{repro_i}  davidson conv={st['converged']} n_sigma={st['n_sigma']} |r|={res:.1e} "
              f"[{time.perf_counter() - t0:.0f} s]", flush=True)
    except Exception as exc:  # noqa: BLE001 - a survey: report and go on
        bad += 1
        print(f"BAD case {case} norb={norb} nelec=({nea},{neb}) {kind} {na} x {nb}: {exc!r}", flush=True)
print(f"fuzz: {ncases} cases, {bad} bad")
