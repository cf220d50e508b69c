#!/usr/bin/env python
"""Benchmark of the fermionic subspace-diagonalization hot path on MI355X.

Contract (see task statement): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is
launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` with one
rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric "Davidson sigma-vectors/sec & wall-clock to E0, N2 (16e,30o) 1e5 dets"):
synthetic N2-sized FCIDUMP integrals (norb = 30, nelec = (8, 8); written to and read back from a FCIDUMP file
before the timed region), particle-conserving random bitstrings, na = nb = 317 strings per spin
(D = 100 489 determinants), ONE independent subspace (subsample batch) per GPU -- weak scaling, batches differ
by seed.

A *step* is one call of the PRODUCT's public entry point on one batch per GPU:
  N = 1: ``qiskit_addon_sqd_amd.fermion.solve_fermion((strs_a, strs_b), hcore, eri)`` -- string checks, CI-string
         link tables + hdiag built on the device, Davidson to pyscf's default tolerance (tol 1e-9) from pyscf's initial
         guess, then <c|H|c>, <S^2>, orbital occupancies, and the amplitude matrix returned to the host as an
         ``SCIState``: everything the reference's ``solve_fermion`` returns, Python layer included;
  N > 1: ``qiskit_addon_sqd_amd.distributed.solve_sci_batch_distributed(batches, ...)`` with N batches -- rank r solves
         batch r, then the path's only exchange: ONE all-reduce of the (E, occ_a, occ_b) records over RCCL, argmin
         (reference semantics, fermion.py:577) and the broadcast of the winner's amplitudes.  Nothing is omitted.
Integrals are resident in HBM (the context is created in the warm-up).  ``value`` = sigma-vectors built by all ranks /
max-over-ranks wall time of the K steps.  ``native_ms_per_step`` (N = 1) times the same solve through the C ABI alone.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

F64_MFMA_PEAK_TFLOPS = 78.6  # dense f64 matrix peak of the MI355X (public spec)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PROFILE_ROUND = "r06"


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
    p.add_argument("--skip-secondary", action="store_true", help="skip the secondary entries (HF-centred 317^2, 1e4 x 1e4 sigma)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    p.add_argument("--cpu-threads", type=int, default=0)
    p.add_argument("--extra", action="store_true", help="also measure a D ladder")
    p.add_argument("--time-sigma-every", type=int, default=0,
                   help="bracket every k-th sigma launch of the TIMED REGION with HIP events (a cross-check of the roofline "
                        "leg, `in_region_samples`); default 0 = none: a bracket and the empty bracket behind it cost 10-20 us "
                        "of stream time on a 130 us step -- with every 8th launch bracketed a third of the steps carried "
                        "one and the mean of the steps sat 6 % above their median (round 6).  The roofline leg's duration "
                        "comes from 100 solves right behind the region with EVERY launch bracketed, either way")
    return p.parse_args()


def make_batch(args, seed, strings=None):
    from qiskit_addon_sqd_amd import synthetic as S

    gen = S.uniform_strings if (strings or args.strings) == "uniform" else S.hf_centred_strings
    return gen(args.norb, args.nelec, args.na, seed), gen(args.norb, args.nelec, args.nb, seed + 7919)


def integrals_through_fcidump(args, rank):
    """The synthetic Hamiltonian really travels through the FCIDUMP format: written, read back,
    compared bit for bit -- outside the timed region."""
    from qiskit_addon_sqd_amd import synthetic as S

    h1, eri = S.synthetic_integrals(args.norb)
    with tempfile.TemporaryDirectory() as tmp:
        path = Path(tmp) / f"FCIDUMP_{args.norb}_{rank}"
        S.write_fcidump(path, h1, eri, nelec=2 * args.nelec, ms2=0)
        h1r, erir, nelec, _, _ = S.read_fcidump(path)
    assert nelec == 2 * args.nelec and np.array_equal(h1r, h1) and np.array_equal(erir, eri)
    return h1r, erir


def pmc_traffic(args):
    """HBM bytes per k_sigma launch from the COMMITTED rocprofv3 PMC passes of this workload (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950);
    bench.py cannot collect counters itself.  (None, source) when no matching profile is committed."""
    if (args.norb, args.nelec, args.na, args.nb) != (30, 8, 317, 317):
        return None, None
    for rnd in (PROFILE_ROUND, "r05", "r03", "r01"):
        f = ROOT / "profiles" / rnd / "pmc" / f"final_{args.strings}317_pmc_summary.json"
        try:
            d = json.loads(f.read_text())["HBM_BYTES"]
            # the H-sigma instantiation is the one the Davidson launches (most dispatches); S^2 runs once per solve
            key = max((k for k in d if "k_sigma" in k and "reduce" not in k), key=lambda k: d[k]["dispatches"])
            return d[key]["hbm_bytes_per_launch"], f"committed profile {f.relative_to(ROOT)} (not measured in this run)"
        except Exception:
            continue
    return None, None


def cpu_baseline(args, h1, eri, sa, sb, n_sigma_gpu):
    """The reference's CPU path on this box's host cores, bounded sample of the same workload.  Preferred: pyscf itself
    (``kernel_fixed_space`` as the reference calls it, fermion.py:810-818) when it is importable on the box; otherwise
    oracle O2, the C restatement of pyscf's dense gather/dgemm/scatter formulation (OpenMP over strings + sequential
    OpenBLAS dgemm)."""
    pyscf_note = "pyscf not importable on this box"
    try:
        return cpu_baseline_pyscf(args, h1, eri, sa, sb)
    except ImportError:
        pass
    except Exception as exc:  # pyscf present but the call failed: say so and fall back to the port
        pyscf_note = f"pyscf call failed: {exc!r}"
    from oracle import sci_ref as R

    lib = R.load()
    # physical cores, capped at 64: the OpenBLAS bundled with numpy/scipy keeps per-thread metadata for
    # 64 callers and crashes beyond that when dgemm is entered from more OpenMP threads
    threads = args.cpu_threads or max(1, min(64, (os.cpu_count() or 2) // 2))
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
    # wall clock to E0, MEASURED where it is affordable: the whole reference flow (string check, tables, hdiag, pyscf's
    # Davidson to tol 1e-9 on the restated contract_2e, <c|H|c>, occupancies -- oracle/sci_ref.py: solve_fermion_ref) on
    # the same subspace.  Uniform-random sets converge in 2-3 sigma builds (a few seconds of CPU); for HF-centred sets
    # (~30 builds) the figure stays an estimate and says so.
    measured = None
    if per_sigma * (n_sigma_gpu + 2) < 4.0 * args.cpu_seconds:
        t0 = time.perf_counter()
        e_cpu, _amps, _occ, nsig_cpu = R.solve_fermion_ref((sa, sb), h1, eri)
        measured = {"wall_to_e0_s": time.perf_counter() - t0, "sigma_builds": int(nsig_cpu), "energy": float(e_cpu)}
    return {
        "value": 1.0 / per_sigma,
        "unit": "sigma-vectors/s",
        "cores": threads,
        "kind": "port",
        "sample": (f"{n + 1} sigma builds of the same {prob.na}x{prob.nb} subspace (pyscf dense formulation, "
                   f"{prob.dense_flops_per_sigma():.2e} flop each, {R.blas_name()}); tables+hdiag {t_setup:.2f} s"),
        "s_per_sigma": per_sigma,
        **({"wall_to_e0_s": measured["wall_to_e0_s"], "wall_to_e0_sigma_builds": measured["sigma_builds"],
            "wall_to_e0_energy": measured["energy"],
            "wall_to_e0_note": "measured: oracle/sci_ref.py solve_fermion_ref (the reference's whole flow) run to convergence"}
           if measured else
           {"est_wall_to_e0_s": t_setup + per_sigma * n_sigma_gpu,
            "est_note": "estimate: CPU per-sigma time x the GPU run's sigma count (the CPU Davidson itself is not run: "
                        "it would take minutes at this link density)"}),
        "cores_note": (f"{threads} OpenMP threads of {os.cpu_count()} logical CPUs: the OpenBLAS bundled with numpy keeps "
                       "per-thread state for 64 callers and crashes when dgemm is entered from more OpenMP threads"),
        "pyscf": pyscf_note,
    }


def cpu_baseline_pyscf(args, h1, eri, sa, sb):
    """pyscf itself on the host cores, called exactly as the reference does (fermion.py:803-818); raises ImportError
    when pyscf is absent -- it is absent from the build container, so this leg is opportunistic."""
    import pyscf
    from pyscf import fci, lib

    norb, nelec = args.norb, (args.nelec, args.nelec)
    myci = fci.selected_ci.SelectedCI()
    count = [0]
    inner = myci.contract_2e

    def counted(*a, **k):
        count[0] += 1
        return inner(*a, **k)

    myci.contract_2e = counted
    t0 = time.perf_counter()
    e, _civec = fci.selected_ci.kernel_fixed_space(myci, h1, eri, norb, nelec, ci_strs=(np.asarray(sa), np.asarray(sb)))
    wall = time.perf_counter() - t0
    return {"value": count[0] / wall if count[0] else None, "unit": "sigma-vectors/s", "cores": int(lib.num_threads()),
            "kind": "reference", "sample": f"pyscf {pyscf.__version__} kernel_fixed_space to convergence on the same "
            f"{len(sa)}x{len(sb)} subspace: {count[0]} contract_2e calls in {wall:.2f} s",
            "wall_to_e0_s": wall, "energy": float(e)}


def oracle_energy_check(args, h1, eri, sa, sb, e_gpu):
    """E0 of the benchmarked subspace from the oracle's own Davidson on the string-space operator (outside the
    timed region; tests/test_gpu_parity.py holds the full comparison)."""
    from oracle import sqd_oracle as O

    op = O.StringSpaceOperator(h1, eri, sa, sb, args.norb)
    hd = O.make_hdiag(h1, eri, sa, sb, args.norb).ravel()
    conv, e_ref, _, nsig = O.davidson_pyscf(op, O.init_guess(hd, len(sa), len(sb), (args.nelec, args.nelec)), hd,
                                            tol=1e-11, max_cycle=200)
    return {"oracle_energy": float(e_ref), "abs_diff_ha": abs(float(e_ref) - e_gpu), "oracle_converged": bool(conv),
            "oracle_sigma_builds": int(nsig), "oracle": "O1s string-space sigma + pyscf-flow Davidson (numpy)"}


def roofline_entry(ctx, t_bracket_ms, t_apply_ms, n_timed, traffic=None, source=None, t_empty_ms=0.0):
    """t_bracket_ms: average HIP-event bracket around the sigma kernel; t_empty_ms: average EMPTY bracket recorded right
    behind it (what event records cost by themselves: ~5 us, more than half of a 9 us bracket at batch size).  A kernel
    bracket hides part of that cost behind the kernel's own dispatch, so only a FRACTION of the empty bracket is
    subtracted: 0.6, calibrated in round 6 against rocprofv3 --kernel-trace of this very command
    (profiles/r06/roofline_calibration.txt: 200 bracketed launches, bracket 8.97 us, empty 5.28 us -> 5.80 us; the
    trace's 876 live launches: mean 5.65, median 5.88 us).  Round 2's 0.5 (6.3 us here) sat 7-12 % above the trace;
    subtracting all of it would claim 3.7 us / frac 0.16 at the headline."""
    t_kernel_ms = t_bracket_ms - 0.6 * t_empty_ms
    b_alg = ctx.sigma_bytes()
    b_need = ctx.sigma_bytes_needed()
    ach = b_alg / (t_kernel_ms * 1e-3) / 1e9 if t_kernel_ms > 0 else 0.0
    fused = {}
    if ctx.sigma_kernel() == "k_sigma_direct":
        fused = {"fused_in_run": "inside a Davidson run of fewer than 2e5 determinants the sigma builds that are not being timed "
                                 "go out as sqd::k_sigma_dots_eig -- this kernel's per-element code in the geometry of the "
                                 "dot-product / eigen-step kernel behind it, ONE launch, the same bits; a bracketed launch is the "
                                 "stand-alone sqd::k_sigma_direct, which is what avg_launch_ms, frac and traffic describe"}
    return {
        **fused,
        "bound": "hbm", "kernel": "sqd::" + ctx.sigma_kernel(), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
        "bytes_per_launch": b_alg, "bytes_needed": b_need,
        "frac_on_bytes_needed": (b_need / (t_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_kernel_ms > 0 else 0.0,
        "avg_launch_ms": t_kernel_ms, "event_bracket_ms": t_bracket_ms, "empty_bracket_ms": t_empty_ms,
        "sigma_application_ms": t_apply_ms, "timed_launches": n_timed,
        "note": "bytes_per_launch = SURVEY 8d B_sigma = 16 D + 8 links + 8 (nnorb_s^2 + nnorb_a^2); bytes_needed = what this "
                "formulation must read and write once (vectors, hdiag, the link records and the integral / J rows it "
                "touches).  The working set is cache resident at D <= 1e7, so HBM traffic is far below peak by construction",
    }


def main():
    args = parse_args()
    # stdout carries ONE JSON line.  Libraries loaded below write banners to file descriptor 1 (RCCL prints its version
    # block there when the process group comes up), so everything but that line goes to stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
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

    from qiskit_addon_sqd_amd import distributed as D
    from qiskit_addon_sqd_amd import fermion as F
    from qiskit_addon_sqd_amd.distributed import solve_sci_batch_distributed

    xms: list = []  # per-step wall clock of the exchange on this rank (N > 1)
    h1_rw, eri_rw = integrals_through_fcidump(args, rank)
    # The solver context of a Hamiltonian is found through a hash of the integral tensors.  A WRITEABLE array may have
    # been edited in place since the last call, so the library hashes it in full on every call (~1 ms for the 6.5 MB at
    # norb = 30); read-only arrays are recognised by identity.  A caller that solves many subspaces of one Hamiltonian
    # freezes the tensors once -- the package's own SQD loop does -- and so does this benchmark; the cost with
    # writeable tensors is reported beside the headline (`ms_per_step_writeable_integrals`).
    h1, eri = F.freeze_integrals(h1_rw, eri_rw)
    batches = [make_batch(args, 1000 + r) for r in range(world)]
    sa, sb = batches[rank]
    nelec = (args.nelec, args.nelec)

    import collections

    held = collections.deque(maxlen=64)  # the states of the latest steps (N = 1), read before the clock stops

    def one_step():
        """One call of the product's public API; returns (energy, Davidson statistics of this rank's solve)."""
        if dist is None:
            e, _state, _occ, _s2 = F.solve_fermion((sa, sb), h1, eri, spin_sq=args.spin_sq, device=local_rank)
            # (the call returns with energy, occupancies and <S^2>; the amplitudes may still be landing in the state's
            # page-locked array -- the timed region ends only when every state has been READ on the host, see below)
            held.append(_state)
        else:
            res = solve_sci_batch_distributed(batches, h1, eri, args.norb, nelec, spin_sq=args.spin_sq,
                                              device=local_rank, compute_rdms=False)
            win = min(res, key=lambda r: r.energy)
            e = win.energy
            if rank == 0:  # the control process consumes the winner's state (carry-over, reference fermion.py:608-631)
                assert win.sci_state.amplitudes.shape == (len(win.sci_state.ci_strs_a), len(win.sci_state.ci_strs_b))
            xms.append(D.last_exchange_ms)
        return e, F.last_solve_stats()

    # Untimed device spin-up, then the W warm-up steps.  The first ~0.1 s of GPU activity of a process contains one or
    # two 30-50 ms stalls that have nothing to do with the work submitted (profiles/r02/stall_probe.txt: identical
    # 0.27 ms solves, calls 113 and 189 take 1.2 and 40 ms, then none in the next thousands; HF-centred 3.5 ms solves
    # show none after their first call) -- the device settling into its power state.  A timed region that starts a
    # few milliseconds after the process touched the GPU catches them at random; 0.3 s of the same solves first.
    # (event sampling of the sigma kernel is switched on BEFORE the warm-up: the first sampled solve creates the events)
    # Python's cyclic collector runs a FULL pass every thousand calls or so, and a full pass visits everything
    # `import torch` created: 35 ms -- 170 steps' worth -- landing in the timed region or not by the count of objects
    # allocated so far (measured: a 100-step N > 1 run read 0.22 or 0.59 ms per step).  gc.freeze() BEFORE the spin-up
    # (a 40 ms pause right before the timed region would leave the GPU idle and the first timed step slow) moves what
    # exists -- the imports -- to the permanent generation: collections still run during the timed steps, over what the
    # steps themselves allocate.  SQD_BENCH_GC=default leaves the collector as it is.
    if os.environ.get("SQD_BENCH_GC") != "default":
        import gc

        gc.collect()
        gc.freeze()
    F.set_profiling(time_sigma_every=args.time_sigma_every)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        one_step()
    for _ in range(args.warmup):
        e, st = one_step()
    if os.environ.get("SQD_BENCH_GC") != "default":
        # (again, over what the spin-up allocated -- a millisecond now: the collector's generation counters start the
        # timed region at zero, so only young-generation passes fall into a region of 20 steps)
        gc.collect()
        gc.freeze()

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    sync()
    xms.clear()
    held.clear()
    t0 = time.perf_counter()
    nsig = 0
    ms_sigma = ms_apply = ms_empty = 0.0
    n_timed = 0
    step_marks = []  # (one perf_counter per step: the median of the steps is reported beside the mean)
    for _ in range(args.steps):
        e, st = one_step()
        step_marks.append(time.perf_counter())
        nsig += st["n_sigma"]
        n_timed += st["n_sigma_timed"]
        ms_sigma += st["ms_sigma_kernel"]
        ms_apply += st["ms_sigma"]
        ms_empty += st["ms_event_overhead"]
    for _st in held:  # every state of the timed steps is on the host, and is read, inside the timed region
        assert _st.amplitudes.shape == (len(sa), len(sb))
    sync()
    elapsed = time.perf_counter() - t0
    exchange_ms = float(np.mean(xms)) if xms else None
    if dist is not None and rank == 0:
        # the batched native solve of the N > 1 step does not bracket its sigma launches: the roofline leg's kernel
        # time comes from a few untimed single solves of the same batch behind the timed region
        nsig_keep = nsig
        for _ in range(5):
            F.solve_fermion((sa, sb), h1, eri, spin_sq=args.spin_sq, device=local_rank)
            stx = F.last_solve_stats()
            n_timed += stx["n_sigma_timed"]
            ms_sigma += stx["ms_sigma_kernel"]
            ms_apply += stx["ms_sigma"]
            ms_empty += stx["ms_event_overhead"]
        nsig = nsig_keep
    F.set_profiling(0)
    step_ms = np.diff([t0] + step_marks) * 1e3
    if os.environ.get("SQD_BENCH_DEBUG") and rank == 0:
        d = step_ms
        print("step ms:", " ".join(f"{x:.3f}" for x in d), "| closing sync %.3f" % ((t0 + elapsed - step_marks[-1]) * 1e3),
              file=sys.stderr)
    # device time of the two phases of a solve (HIP events around the table build and around the Davidson run): five
    # extra, untimed steps -- the events are bubbles in the stream and are kept out of the timed region
    ctx_t = F._get_context(h1, eri, local_rank)
    ctx_t.set_phase_timing(True)
    ms_dav = ms_setup = 0.0
    for _ in range(5):
        _, stp = one_step()
        ms_dav += stp["ms_total"] / 5
        ms_setup += stp["ms_setup"] / 5
    ctx_t.set_phase_timing(False)

    tot = torch.tensor([float(nsig), elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        nsig_all, elapsed_max = float(tot[0].item()), float(tmax[1].item())
    else:
        nsig_all, elapsed_max = float(nsig), elapsed

    if rank == 0:
        ctx = F._get_context(h1, eri, local_rank)  # the context the timed solves used (cache hit)
        t_sigma_ms = ms_sigma / max(n_timed, 1)
        # The roofline leg's launch duration: 100 more solves BEHIND the timed region with EVERY sigma launch inside its own
        # HIP-event bracket and an empty bracket behind it (>= 200 samples under the conditions of the timed solves: new
        # vectors, the Davidson's own launches around them; round 5's line carried 5 samples taken inside the region and
        # swung 0.060 <-> 0.096 from box to box).  The in-region samples stay as a cross-check.
        F.set_profiling(time_sigma_every=1)
        for _ in range(5):  # (not counted: the first bracketed launches of a process pay for the events' own set-up)
            F.solve_fermion((sa, sb), h1, eri, spin_sq=args.spin_sq, device=local_rank)
        pr_n = 0
        pr_k = pr_a = pr_e = 0.0
        for _ in range(100):
            F.solve_fermion((sa, sb), h1, eri, spin_sq=args.spin_sq, device=local_rank)
            stx = F.last_solve_stats()
            pr_n += stx["n_sigma_timed"]
            pr_k += stx["ms_sigma_kernel"]
            pr_a += stx["ms_sigma"]
            pr_e += stx["ms_event_overhead"]
        F.set_profiling(0)
        br = {"kernel_ms": pr_k / max(pr_n, 1), "apply_ms": pr_a / max(pr_n, 1), "empty_ms": pr_e / max(pr_n, 1), "launches": pr_n}
        ns_a, nd_a = ctx.link_counts(0)
        ns_b, nd_b = ctx.link_counts(1)
        traffic, source = pmc_traffic(args)
        out = {
            "metric": "Davidson sigma-vectors/sec (complete solve_fermion through the Python API: string checks + tables + "
                      "Davidson to tol 1e-9 + observables + SCIState)",
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
                "entry_point": ("qiskit_addon_sqd_amd.fermion.solve_fermion" if dist is None else
                                "qiskit_addon_sqd_amd.distributed.solve_sci_batch_distributed"),
                "parallelism": f"batch-per-gpu x{world}" + (" + all_reduce(raw observables records) on the solver's stream -> "
                                                            "argmin; winner's state to rank 0 only, read there in the "
                                                            "timed step" if dist is not None else ""),
            },
            "wall_to_e0_ms": 1e3 * elapsed_max / args.steps,
            "exchange_ms": exchange_ms,
            "sigma_per_solve": nsig / args.steps,
            "value_note": ("sigma builds of the solves / wall clock of the WHOLE steps: the denominator holds the table build, "
                           "the BLAS-1 part of every iteration, the observables, the state's trip to the host and the Python "
                           "layer, not the sigma kernel alone (that is `roofline`)"),
            "davidson_ms_per_solve": ms_dav,
            "tables_ms_per_solve": ms_setup,
            "energy": float(e), "converged": int(st["converged"]), "residual": float(st["residual"]),
            "links": {"alpha_single": ns_a, "alpha_double": nd_a, "beta_single": ns_b, "beta_double": nd_b},
            "roofline": roofline_entry(ctx, br["kernel_ms"], br["apply_ms"], br["launches"], traffic, source, br["empty_ms"]),
            "ms_per_step_median": float(np.median(step_ms)),
            "ms_per_step_min": float(np.min(step_ms)),
        }
        out["roofline"]["bracket_source"] = ("100 solves behind the timed region, every sigma launch in its own HIP-event bracket + an "
                                             "empty bracket behind it")
        out["roofline"]["in_region_samples"] = {"launches": n_timed, "event_bracket_ms": t_sigma_ms,
                                                "empty_bracket_ms": ms_empty / max(n_timed, 1)}
        # SURVEY 8d's second unit: one Davidson ITERATION, B_iter = B_sigma + 8 D (4 m + 6) at basis size m (the mean
        # m of this solve: sigma builds 1..n), over the device time of the Davidson run per sigma build
        n_it = max(nsig / args.steps, 1.0)
        m_mean = 0.5 * (n_it + 1.0) if n_it <= 12 else 6.5
        b_iter = ctx.sigma_bytes() + 8.0 * args.na * args.nb * (4.0 * m_mean + 6.0)
        t_iter_ms = ms_dav / n_it if ms_dav > 0 else 0.0
        out["roofline_iter"] = {
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "bytes_per_iteration": b_iter, "mean_basis_size": m_mean,
            "ms_per_iteration": t_iter_ms,
            "achieved": (b_iter / (t_iter_ms * 1e-3) / 1e9) if t_iter_ms > 0 else None,
            "frac": (b_iter / (t_iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_iter_ms > 0 else None,
            "note": "B_iter = B_sigma + 8 D (4 m + 6) (SURVEY 8d); time = HIP events around the Davidson run / sigma builds "
                    "(five untimed extra steps with phase timing on)",
        }
        if world == 1:
            out["native_ms_per_step"] = native_step_ms(ctx, sa, sb, args)
            def _loop(hh, ee, n):  # the timed region's shape: states held, every one read before the clock stops
                keep = []
                t_ = time.perf_counter()
                for _ in range(n):
                    keep.append(F.solve_fermion((sa, sb), hh, ee, spin_sq=args.spin_sq, device=local_rank)[1])
                for st_ in keep:
                    assert st_.amplitudes.shape == (len(sa), len(sb))
                return 1e3 * (time.perf_counter() - t_) / n

            _loop(h1_rw, eri_rw, 3)
            t_rw = min(_loop(h1_rw, eri_rw, 40), _loop(h1_rw, eri_rw, 40))
            t_fr = min(_loop(h1, eri, 40), _loop(h1, eri, 40))
            out["ms_per_step_writeable_integrals"] = t_rw
            out["ms_per_step_frozen_same_loop"] = t_fr
            out["writeable_over_frozen"] = t_rw / t_fr
            out["writeable_integrals_note"] = ("plain (writeable) numpy tensors, the reference user's call: the previous call's "
                                               "solver context is taken at once and the library's hash threads digest the tensors "
                                               "while the solve runs (sqd_hash_start / sqd_hash_finish); a mismatch repeats the solve "
                                               "(fermion._run_on_context); `ms_per_step_frozen_same_loop`: the frozen call timed by the same "
                                               "40-step loop right behind it (best of two runs each)")
            if args.spin_sq is not None:
                # the oracle's Davidson here runs the bare operator; the penalised solve is compared with the
                # reference flow in tests/test_gpu_parity.py (spin-penalty cases), not in the bench
                out["energy_check"] = {"skipped": "spin penalty on: the bench's oracle check runs the bare operator"}
            else:
                try:
                    out["energy_check"] = oracle_energy_check(args, h1, eri, sa, sb, float(e))
                except Exception as exc:  # never take the GPU number down
                    out["energy_check"] = {"error": repr(exc)}
        # (the secondary GPU entries first: the CPU baseline leaves 64 OpenMP workers spinning behind it)
        if world == 1 and not args.skip_secondary:
            out["secondary"] = secondary_entries(args, h1, eri, local_rank)
        if world == 1 and not args.skip_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(args, h1, eri, sa, sb, nsig / args.steps)
            except Exception as exc:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "sigma-vectors/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {exc!r}"}
        if args.extra and world == 1:
            out["extra"] = extra_measurements(args, ctx)
        print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def native_step_ms(ctx, sa, sb, args, steps=20):
    """The same solve through the C ABI alone (sqd_solve_strings), Python layer excluded."""
    for _ in range(3):
        ctx.solve(sa, sb, spin_sq=args.spin_sq, shift=0.1)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.solve(sa, sb, spin_sq=args.spin_sq, shift=0.1)
    return 1e3 * (time.perf_counter() - t0) / steps


def secondary_entries(args, h1, eri, device):
    """Named secondary lines of every N = 1 run (not the headline): the HF-centred variant of the headline size -- the
    workload that actually exercises the coupling enumeration -- and one sigma launch at the only HBM-relevant size,
    uniform 1e4 x 1e4 (D = 1e8, 0.8 GB per vector)."""
    from qiskit_addon_sqd_amd import fermion as F
    from qiskit_addon_sqd_amd import synthetic as S

    res = {}
    if (args.norb, args.nelec) != (30, 8):
        return res
    ctx = F._get_context(h1, eri, device)
    # --- HF-centred 317 x 317 through the Python API
    sa, sb = S.hf_centred_strings(30, 8, 317, 1001), S.hf_centred_strings(30, 8, 317, 1001 + 7919)
    # (device spin-up as in main(): the GPU has been idle through the host-side checks, and the first ~0.1 s of activity
    # after idle holds one or two 30-50 ms stalls that have nothing to do with the work submitted)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        F.solve_fermion((sa, sb), h1, eri, device=device)
    F.set_profiling(8)
    steps, nsig, ms_k, ms_a, nt, ms_dav, ms_e = 10, 0, 0.0, 0.0, 0, 0.0, 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        e, *_ = F.solve_fermion((sa, sb), h1, eri, device=device)
        st = F.last_solve_stats()
        nsig += st["n_sigma"]; ms_k += st["ms_sigma_kernel"]; ms_a += st["ms_sigma"]; nt += st["n_sigma_timed"]
        ms_e += st["ms_event_overhead"]
    dt = time.perf_counter() - t0
    F.set_profiling(0)
    ctx.set_phase_timing(True)
    for _ in range(3):
        F.solve_fermion((sa, sb), h1, eri, device=device)
        ms_dav += F.last_solve_stats()["ms_total"] / 3
    ctx.set_phase_timing(False)
    res["hf_centred_317x317"] = {
        "ms_per_solve": 1e3 * dt / steps, "sigma_per_solve": nsig / steps, "sigma_vectors_per_s": nsig / dt,
        "us_per_davidson_iteration": 1e3 * ms_dav / max(nsig / steps, 1), "energy": float(e),
        "roofline": roofline_entry(ctx, ms_k / max(nt, 1), ms_a / max(nt, 1), nt, t_empty_ms=ms_e / max(nt, 1)),
    }
    # --- MFMA roofline of the dense same-spin product of that subspace (north_star: "MFMA utilisation against the chip's
    # roofline"): the kernel alone, HIP events around 20 launches, as one problem per launch (what a single solve
    # launches: 25 tiles x 8 k-ranges, latency-bound) and as 16 per launch (a batched solve's full chip)
    try:
        ctx.set_subspace(sa, sb)
        if ctx.sigma_kernel().startswith("k_same_spin_mfma"):
            for copies, key in ((1, "roofline_mfma"), (16, "roofline_mfma_16_per_launch")):
                ms1, fl1 = ctx.time_dense(20, copies)
                res["hf_centred_317x317"][key] = {
                    "bound": "mfma", "kernel": "sqd::k_same_spin_mfma_b", "dtype": "f64", "problems_per_launch": copies,
                    "flops_per_launch": fl1, "avg_launch_ms": ms1, "achieved": fl1 / (ms1 * 1e-3) / 1e12,
                    "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl1 / (ms1 * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                    "note": "G = H_a C + C H_b on the zero-padded orders (320): 2 pa^2 pb + 2 pa pb^2 flops per problem; "
                            "v_mfma_f64_16x16x4_f64; peak = dense f64 matrix rate (public spec, SURVEY 8d); counters: "
                            "profiles/r05/pmc/final_mfma_batch_hf16_summary.json",
                }
    except Exception as exc:
        res["hf_centred_317x317"]["roofline_mfma"] = {"error": repr(exc)}
    # --- the same HF-centred solve with the TIGHT residual rule (|r| < sqrt(tol)/32, this library's default up to round 4
    # and still its rule with a spin penalty) next to the default above (pyscf's |r| < sqrt(tol))
    try:
        e_def = float(e)
        kw = {"tol_residual": 1e-9 ** 0.5 / 32.0}
        for _ in range(2):
            F.solve_fermion((sa, sb), h1, eri, device=device, **kw)
        t0 = time.perf_counter()
        nsig2 = 0
        for _ in range(steps):
            e2, *_ = F.solve_fermion((sa, sb), h1, eri, device=device, **kw)
            nsig2 += F.last_solve_stats()["n_sigma"]
        dt2 = time.perf_counter() - t0
        res["hf_centred_317x317_tight_residual_rule"] = {
            "ms_per_solve": 1e3 * dt2 / steps, "sigma_per_solve": nsig2 / steps, "energy": float(e2),
            "abs_diff_to_default_rule_ha": abs(float(e2) - e_def),
            "note": "tol_residual = sqrt(tol)/32: occupancies at 1e-6 instead of 1e-4; the default rule is the reference solver's",
        }
    except Exception as exc:
        res["hf_centred_317x317_tight_residual_rule"] = {"error": repr(exc)}
    # --- BASELINE config 3 on one GPU: the whole ci_strings list of one SQD iteration through solve_sci_batch (the
    # sci_solver seam, reference fermion.py:432): 8 uniform subsample batches, and 16 HF-centred ones
    for key, gen, nbt in (("config3_8_batches_one_gpu", S.uniform_strings, 8), ("hf_centred_16_batches_one_gpu", S.hf_centred_strings, 16)):
        try:
            bl = [(gen(30, 8, 317, 100 + i), gen(30, 8, 317, 900 + i)) for i in range(nbt)]
            entry = {"batches": nbt, "na": 317, "nb": 317}
            for mode, kw in (("batched", {}), ("one_by_one", {"concurrency": 1})):
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.2:
                    F.solve_sci_batch(bl, h1, eri, 30, (8, 8), compute_rdms=False, **kw)
                ts = []
                for _ in range(7):
                    t0 = time.perf_counter()
                    out_b = F.solve_sci_batch(bl, h1, eri, 30, (8, 8), compute_rdms=False, **kw)
                    ts.append(time.perf_counter() - t0)
                entry[mode + "_ms_per_batch"] = 1e3 * float(np.median(ts)) / nbt
                if mode == "batched":
                    nsig_b = sum(st["n_sigma"] for st in F._TLS.batch_stats)
                    entry["sigma_builds"] = nsig_b
                    entry["sigma_vectors_per_s"] = nsig_b / float(np.median(ts))
                    entry["lowest_energy"] = float(min(r.energy for r in out_b))
            entry["speedup"] = entry["one_by_one_ms_per_batch"] / entry["batched_ms_per_batch"]
            entry["note"] = ("solve_sci_batch: ONE native call (sqd_solve_batch), every launch advances all batches; only the "
                             "lowest-energy state is brought to the host, the others on access; results bit-identical to one by one")
            res[key] = entry
        except Exception as exc:
            res[key] = {"error": repr(exc)}
    # --- CONNECTED subspaces at D = 1e6, 9e6 and 2.5e7 (HF-centred 1000^2, 3000^2, 5000^2): the regime the reference advertises
    # (README.md:78, "subspace dimensions of ~1e7") -- sigma against the HBM roofline, the whole Davidson iteration, and
    # the dense same-spin product of the same subspace on the matrix cores (orders 1024 / 3072) against the f64 MFMA peak
    for n in (1000, 3000, 5000):
        key = f"hf_centred_{n}x{n}"
        try:
            sa, sb = S.hf_centred_strings(30, 8, n, 11), S.hf_centred_strings(30, 8, n, 13)
            ctx.set_subspace(sa, sb)
            ctx.time_sigma(2)
            t_sig = ctx.time_sigma(5)
            entry = {"D": n * n, "links": [ctx.link_counts(0), ctx.link_counts(1)],
                     "roofline": roofline_entry(ctx, t_sig, t_sig, 5)}
            entry["roofline"]["note_kernels"] = (
                "one sigma = the same-spin product (sqd::k_spmm_grouped between two k_spmm_transpose launches) + the "
                "opposite-spin part and the diagonal by whole rows, the beta link list in registers (sqd::k_opp_rows up to "
                "3072 columns; sqd::k_opp_src, passes over ranges of the source column, beyond); avg_launch_ms is the whole "
                "application (HIP events around 5 of them); per-kernel times: profiles/r06/final_hf1000_kernel_stats.csv, "
                "final_hf3000_kernel_stats.csv, final_hf5000_kernel_stats.csv")
            ctx.davidson(fetch=False)  # (first call at this size grows the arenas: not timed)
            ctx.sync()
            t0 = time.perf_counter()
            _, st_c = ctx.davidson(fetch=False)
            ctx.sync()
            ms_c = 1e3 * (time.perf_counter() - t0)
            ns_c = max(int(st_c["n_sigma"]), 1)
            m_mean = 0.5 * (ns_c + 1.0) if ns_c <= 12 else 6.5
            b_iter = ctx.sigma_bytes() + 8.0 * n * n * (4.0 * m_mean + 6.0)
            entry["davidson"] = {"ms_per_solve": ms_c, "sigma_builds": ns_c, "converged": int(st_c["converged"]),
                                 "energy": float(st_c["e_davidson"]), "us_per_iteration": 1e3 * ms_c / ns_c,
                                 "residual_rule": "pyscf's |r| < sqrt(tol) (the default without a spin penalty)"}
            entry["roofline_iter"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "bytes_per_iteration": b_iter,
                                      "mean_basis_size": m_mean, "ms_per_iteration": ms_c / ns_c,
                                      "achieved": b_iter / (ms_c / ns_c * 1e-3) / 1e9,
                                      "frac": b_iter / (ms_c / ns_c * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # the matrix-core formulation of the same-spin part on the same subspace (forced): MFMA roofline at this order
            os.environ["SQD_SIGMA_DENSE"] = "1"
            try:
                ctx.set_subspace(sa, sb)
                if ctx.sigma_kernel().startswith("k_same_spin_mfma"):
                    ms1, fl1 = ctx.time_dense(5, 1)
                    entry["roofline_mfma"] = {
                        "bound": "mfma", "kernel": "sqd::k_same_spin_mfma_b", "dtype": "f64", "problems_per_launch": 1,
                        "flops_per_launch": fl1, "avg_launch_ms": ms1, "achieved": fl1 / (ms1 * 1e-3) / 1e12,
                        "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl1 / (ms1 * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                        "note": f"G = H_a C + C H_b on the zero-padded orders ({(n + 63) // 64 * 64}): 2 pa^2 pb + 2 pa pb^2 flops; the "
                                "blocks are 11 % (1000) / 5.6 % (3000) dense at these sizes, which is why the default same-spin "
                                "formulation from ~1000 strings per spin is the sparse product, not this kernel"}
                    entry["sigma_ms_with_matrix_cores"] = ctx.time_sigma(3)
            finally:
                os.environ.pop("SQD_SIGMA_DENSE", None)
            res[key] = entry
        except Exception as exc:
            res[key] = {"error": repr(exc)}
    # --- one sigma at uniform 1e4 x 1e4
    try:
        n = 10000
        sa, sb = S.uniform_strings(30, 8, n, 11), S.uniform_strings(30, 8, n, 13)
        ctx.set_subspace(sa, sb)
        t_sig = ctx.time_sigma(5)
        res["sigma_uniform_1e4x1e4"] = {"D": n * n, "links": [ctx.link_counts(0), ctx.link_counts(1)],
                                        "roofline": roofline_entry(ctx, t_sig, t_sig, 5)}
        if ctx.sigma_kernel() == "k_sigma_lists":
            res["sigma_uniform_1e4x1e4"]["roofline"]["launches_per_sigma"] = 3
            res["sigma_uniform_1e4x1e4"]["roofline"]["note_kernels"] = (
                "one sigma = k_lists_t4 (single x single term of the strings that have single links) + k_sigma_lists<1> (diagonal + "
                "beta lists on C, rows through LDS) + k_alpha_rows (alpha lists by rows on C, added onto it); avg_launch_ms is the "
                "whole application (HIP events around 5 of them); per-kernel times and counters: "
                "profiles/r05/final_lists_passes_probe.txt, profiles/r04/pmc/final_lists_uniform10000_counters.txt")
        # the whole Davidson solve at D = 1e8 (26 resident vectors of 0.8 GB): what an iteration costs beside its sigma
        ctx.davidson(fetch=False)  # (first call at this size grows the arenas: not timed)
        ctx.sync()
        t0 = time.perf_counter()
        _, st_big = ctx.davidson(fetch=False)
        ctx.sync()
        ms_big = 1e3 * (time.perf_counter() - t0)
        nsb = max(int(st_big["n_sigma"]), 1)
        # B_iter = B_sigma + 8 D (4 m + 6), m = 1 .. n_sigma in a run without restart
        bytes_blas1 = sum(8.0 * n * n * (4.0 * m + 6.0) for m in range(1, nsb + 1))
        ms_blas1 = ms_big - nsb * t_sig
        res["sigma_uniform_1e4x1e4"]["davidson"] = {
            "ms_per_solve": ms_big, "sigma_builds": nsb, "converged": int(st_big["converged"]),
            "ms_per_iteration": ms_big / nsb, "ms_sigma_share": t_sig,
            "blas1_gbs": bytes_blas1 / (ms_blas1 * 1e-3) / 1e9 if ms_blas1 > 0 else None,
            "blas1_frac_of_hbm_peak": bytes_blas1 / (ms_blas1 * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_blas1 > 0 else None,
            "note": "one whole Davidson solve (tol 1e-9) on the resident subspace, state left on the device; "
                    "blas1 = (wall clock - sigma builds x the sigma time above) against SURVEY 8d's 8 D (4 m + 6) bytes per iteration"}
        m_big = 0.5 * (nsb + 1.0) if nsb <= 12 else 6.5
        b_iter_big = ctx.sigma_bytes() + 8.0 * n * n * (4.0 * m_big + 6.0)
        res["sigma_uniform_1e4x1e4"]["roofline_iter"] = {
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "bytes_per_iteration": b_iter_big, "mean_basis_size": m_big,
            "ms_per_iteration": ms_big / nsb, "achieved": b_iter_big / (ms_big / nsb * 1e-3) / 1e9,
            "frac": b_iter_big / (ms_big / nsb * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "B_iter = B_sigma + 8 D (4 m + 6) (SURVEY 8d) over the measured time of one whole iteration"}
        ctx.set_subspace(sa[:16], sb[:16])  # release nothing, but leave a small subspace behind
    except Exception as exc:
        res.setdefault("sigma_uniform_1e4x1e4", {})["error"] = repr(exc)
    return res


def extra_measurements(args, ctx):
    """Secondary numbers (not the headline): a D ladder, both generators."""
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
