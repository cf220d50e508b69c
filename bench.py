#!/usr/bin/env python
"""Benchmark of the fermionic subspace-diagonalization hot path on MI355X.

Contract (see task statement): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is
launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` with one
rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric "Davidson sigma-vectors/sec & wall-clock to E0, N2 (16e,30o) 1e5 dets"):
synthetic N2-sized FCIDUMP integrals (norb = 30, nelec = (8, 8)), particle-conserving random
bitstrings, na = nb = 317 strings per spin (D = 100 489 determinants), ONE independent subspace
(subsample batch) per GPU -- weak scaling, batches differ by seed.

A *step* is one complete native ``solve_fermion`` on one batch: CI-string link tables + hdiag built on
the device from the string lists, Davidson to pyscf's default tolerance (tol 1e-9) from pyscf's
initial guess, then <c|H|c>, orbital occupancies (rdm1 diagonals), <S^2>, and the amplitude matrix
returned to the host.  Integrals are resident in HBM (context created before the timed region).
For N > 1 every step ends with the path's only exchange: one all-reduce of the (E, occ_a, occ_b)
records over RCCL and an argmin (reference semantics, fermion.py:577).  ``value`` = sigma-vectors built by all ranks /
max-over-ranks wall time of the K steps.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--norb", type=int, default=30)
    p.add_argument("--nelec", type=int, default=8, help="electrons per spin")
    p.add_argument("--na", type=int, default=317)
    p.add_argument("--nb", type=int, default=317)
    p.add_argument("--strings", choices=["uniform", "hf"], default="uniform",
                   help="uniform = random particle-conserving bitstrings (BASELINE config); hf = HF-centred")
    p.add_argument("--spin-sq", type=float, default=None)
    p.add_argument("--skip-cpu", action="store_true", help="skip the cpu_baseline leg")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    p.add_argument("--cpu-threads", type=int, default=0)
    p.add_argument("--extra", action="store_true", help="also measure the HF-centred variant and a D ladder")
    p.add_argument("--time-sigma-every", type=int, default=8,
                   help="bracket every k-th sigma launch of the timed region with HIP events (roofline leg); an "
                        "event pair costs ~10 us of stream time, hence sampling")
    return p.parse_args()


def make_batch(args, seed):
    from qiskit_addon_sqd_amd import synthetic as S

    gen = S.uniform_strings if args.strings == "uniform" else S.hf_centred_strings
    return gen(args.norb, args.nelec, args.na, seed), gen(args.norb, args.nelec, args.nb, seed + 7919)


def one_step(ctx, sa, sb, spin_sq, time_every=0):
    """Native body of solve_fermion (qiskit_addon_sqd_amd/fermion.py) on a resident Hamiltonian."""
    ctx.set_subspace(sa, sb)
    amps, st, (e, s2, occ_a, occ_b) = ctx.davidson(spin_sq=spin_sq, shift=0.1, time_sigma_every=time_every,
                                                   observables=True)
    return e, occ_a, occ_b, s2, st, amps


def pmc_traffic_bytes(args):
    """HBM bytes per k_sigma launch from the committed rocprofv3 PMC passes of THIS workload (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950);
    bench.py cannot collect counters itself.  None when no matching profile is committed."""
    if (args.norb, args.nelec, args.na, args.nb) != (30, 8, 317, 317):
        return None
    f = ROOT / "profiles" / "r01" / "pmc" / f"final_{args.strings}317_pmc_summary.json"
    try:
        d = json.loads(f.read_text())["HBM_BYTES"]
        # the H-sigma instantiation is the one the Davidson launches (most dispatches); S^2 runs once per solve
        key = max((k for k in d if "k_sigma<" in k), key=lambda k: d[k]["dispatches"])
        return d[key]["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(args, h1, eri, sa, sb, n_sigma_gpu):
    """Reference-algorithm port (oracle O2: pyscf's dense gather/dgemm/scatter formulation, OpenMP over
    strings + sequential OpenBLAS dgemm) on this box's host cores; bounded sample of the same workload."""
    from oracle import sci_ref as R

    lib = R.load()
    # physical cores, capped at 64: the OpenBLAS bundled with numpy/scipy keeps per-thread metadata for
    # 64 callers and crashes beyond that when dgemm is entered from more OpenMP threads
    threads = args.cpu_threads or max(1, min(64, (os.cpu_count() or 2) // 2))
    if hasattr(lib, "ref_set_threads"):
        lib.ref_set_threads(threads)
    threads = R.num_threads()
    t0 = time.perf_counter()
    prob = R.RefProblem(h1, eri, sa, sb)
    t_setup = time.perf_counter() - t0
    c = np.zeros(prob.na * prob.nb)
    c[np.argmin(prob.hdiag)] = 1.0
    c += 1e-3 * np.random.default_rng(0).standard_normal(c.size)
    t0 = time.perf_counter()
    prob.contract_2e(c)
    t1 = time.perf_counter() - t0
    n = int(min(20, max(1, round(args.cpu_seconds / max(t1, 1e-3)) - 1)))
    t0 = time.perf_counter()
    for _ in range(n):
        prob.contract_2e(c)
    tn = time.perf_counter() - t0
    per_sigma = (t1 + tn) / (n + 1)
    return {
        "value": 1.0 / per_sigma,
        "unit": "sigma-vectors/s",
        "cores": threads,
        "kind": "port",
        "sample": (f"{n + 1} sigma builds of the same {prob.na}x{prob.nb} subspace (pyscf dense formulation, "
                   f"{prob.dense_flops_per_sigma():.2e} flop each, {R.blas_name()}); tables+hdiag {t_setup:.2f} s"),
        "s_per_sigma": per_sigma,
        "est_wall_to_e0_s": t_setup + per_sigma * n_sigma_gpu,
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    # SQD_BENCH_FORCE_DIST=1 (under torchrun with one process): take the N > 1 code path on one GPU, to
    # measure what the per-step exchange costs
    if world > 1 or (os.environ.get("SQD_BENCH_FORCE_DIST") and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from qiskit_addon_sqd_amd import _capi
    from qiskit_addon_sqd_amd import synthetic as S

    h1, eri = S.synthetic_integrals(args.norb)
    sa, sb = make_batch(args, 1000 + rank)
    ctx = _capi.Context(h1, eri, device=local_rank)

    width = 1 + 2 * args.norb
    if dist is not None:
        # buffers of the per-step exchange, allocated once: device table, pinned host record / table
        allrec = torch.zeros((world, width), device=dev, dtype=torch.float64)
        h_rec = torch.zeros(width, dtype=torch.float64).pin_memory()
        h_all = torch.zeros((world, width), dtype=torch.float64).pin_memory()
        # solver and exchange share ONE stream (sqd_ctx_use_stream): waking a second hardware queue per step
        # costs more than the collective itself (36 vs 77-110 us, profiles/probes/_exchange_probe.py)
        xstream = torch.cuda.Stream(device=dev)
        ctx.use_stream(xstream.cuda_stream)
        my_row, rec, table = allrec[rank], h_rec.numpy(), h_all.numpy()  # views, made once

    def exchange(e, oa, ob):
        if dist is None:
            return e, oa, ob
        # same exchange as qiskit_addon_sqd_amd.distributed: ONE all-reduce(sum) of a table whose rows are
        # zero except the owner's record [E, occ_a, occ_b]  (61 doubles per batch at norb = 30); one host
        # synchronisation per step (after the table is back in pinned memory), argmin on the host
        rec[0] = e
        rec[1 : 1 + args.norb] = oa
        rec[1 + args.norb :] = ob
        with torch.cuda.stream(xstream):
            allrec.zero_()
            my_row.copy_(h_rec, non_blocking=True)
            dist.all_reduce(allrec, op=dist.ReduceOp.SUM)
            h_all.copy_(allrec, non_blocking=True)
        xstream.synchronize()
        row = table[int(np.argmin(table[:, 0]))].copy()
        return row[0], row[1 : 1 + args.norb], row[1 + args.norb :]

    for _ in range(args.warmup):
        e, oa, ob, s2, st, _ = one_step(ctx, sa, sb, args.spin_sq)
        exchange(e, oa, ob)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    sync()
    t0 = time.perf_counter()
    nsig = 0
    ms_sigma = 0.0  # k_sigma launches alone (event before .. event right after the kernel)
    ms_apply = 0.0  # whole sigma applications (k_sigma + k_sigma_reduce)
    ms_dav = 0.0
    ms_setup = 0.0
    n_timed = 0
    s_exchange = 0.0  # host time inside the per-step record exchange (N > 1 only)
    for _ in range(args.steps):
        e, oa, ob, s2, st, _ = one_step(ctx, sa, sb, args.spin_sq, args.time_sigma_every)
        tx = time.perf_counter()
        e_best, _, _ = exchange(e, oa, ob)
        s_exchange += time.perf_counter() - tx
        nsig += st["n_sigma"]
        n_timed += st["n_sigma_timed"]
        ms_sigma += st["ms_sigma_kernel"]
        ms_apply += st["ms_sigma"]
        ms_dav += st["ms_total"]
        ms_setup += st["ms_setup"]
    sync()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([float(nsig), elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        nsig_all, elapsed_max = float(tot[0].item()), float(tmax[1].item())
    else:
        nsig_all, elapsed_max = float(nsig), elapsed

    if rank == 0:
        bytes_sigma = ctx.sigma_bytes()
        t_sigma_ms = ms_sigma / max(n_timed, 1)
        achieved = bytes_sigma / (t_sigma_ms * 1e-3) / 1e9 if t_sigma_ms > 0 else 0.0
        ns_a, nd_a = ctx.link_counts(0)
        ns_b, nd_b = ctx.link_counts(1)
        out = {
            "metric": "Davidson sigma-vectors/sec (complete solve_fermion: tables + Davidson to tol 1e-9 + observables)",
            "value": nsig_all / elapsed_max,
            "unit": "sigma-vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"N2-sized synthetic FCIDUMP ({2 * args.nelec}e,{args.norb}o), {args.strings} random "
                             f"particle-conserving bitstrings, na=nb={args.na}x{args.nb} (D={args.na * args.nb}) "
                             f"determinants, 1 subsample batch per GPU"),
                "norb": args.norb, "nelec": [args.nelec, args.nelec], "na": args.na, "nb": args.nb,
                "strings": args.strings, "spin_sq": args.spin_sq,
                "parallelism": f"batch-per-gpu x{world}" + (" + all_reduce(E,occ)->argmin" if world > 1 else ""),
            },
            "wall_to_e0_ms": 1e3 * elapsed_max / args.steps,
            "exchange_ms_per_step": 1e3 * s_exchange / args.steps,  # rank 0's host time in the all-reduce step
            "sigma_per_solve": nsig / args.steps,
            "davidson_ms_per_solve": ms_dav / args.steps,
            "tables_ms_per_solve": ms_setup / args.steps,
            "energy": float(e), "converged": int(st["converged"]), "residual": float(st["residual"]),
            "links": {"alpha_single": ns_a, "alpha_double": nd_a, "beta_single": ns_b, "beta_double": nd_b},
            "roofline": {
                "bound": "hbm",
                "kernel": "sqd::k_sigma",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic_bytes(args),
                "bytes_per_launch": bytes_sigma,
                "avg_launch_ms": t_sigma_ms,  # k_sigma alone, as in the committed rocprofv3 --stats summary
                "sigma_application_ms": ms_apply / max(n_timed, 1),  # incl. k_sigma_reduce (fixed-order row sums)
                "timed_launches": n_timed,  # every --time-sigma-every-th sigma of the timed region (HIP events)
                "note": "algorithmic bytes = 16 D + 8 links + 8 (nnorb_s^2 + nnorb_a^2) (SURVEY 8d); working set is "
                        "cache resident at this D, so HBM traffic is far below peak by construction",
            },
        }
        if world == 1 and not args.skip_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(args, h1, eri, sa, sb, nsig / args.steps)
            except Exception as exc:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "sigma-vectors/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {exc!r}"}
        if args.extra and world == 1:
            out["extra"] = extra_measurements(args, ctx)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def extra_measurements(args, ctx):
    """Secondary numbers (not the headline): HF-centred strings at the same size and a D ladder."""
    from qiskit_addon_sqd_amd import synthetic as S

    res = {}
    # SURVEY 8(d) ladder D = 1e4 (100^2) .. 1e8 (10^4 x 10^4, 26 resident vectors of 0.8 GB), both generators
    for name, gen, n in (("uniform_100", S.uniform_strings, 100), ("hf_100", S.hf_centred_strings, 100),
                         ("hf_317", S.hf_centred_strings, 317), ("uniform_1000", S.uniform_strings, 1000),
                         ("hf_1000", S.hf_centred_strings, 1000), ("uniform_4000", S.uniform_strings, 4000),
                         ("uniform_10000", S.uniform_strings, 10000)):
        sa, sb = gen(args.norb, args.nelec, n, 11), gen(args.norb, args.nelec, n, 13)
        ctx.set_subspace(sa, sb)  # first call at a new size grows the arenas (hipMalloc): not timed
        ctx.davidson(fetch=False)
        t0 = time.perf_counter()
        ctx.set_subspace(sa, sb)
        t_tab = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, st = ctx.davidson(fetch=False)
        t_dav = time.perf_counter() - t0
        t_sig = ctx.time_sigma(5)
        b = ctx.sigma_bytes()
        res[name] = {"D": n * n, "tables_ms": 1e3 * t_tab, "davidson_ms": 1e3 * t_dav, "n_sigma": st["n_sigma"],
                     "sigma_ms": t_sig, "sigma_GBs": b / (t_sig * 1e-3) / 1e9, "converged": st["converged"],
                     "links": [ctx.link_counts(0), ctx.link_counts(1)]}
    return res


if __name__ == "__main__":
    main()
