#!/usr/bin/env python
"""Turns gpurun_out/<round>/ (written by profiles/collect.sh on the GPU box) into the tracked summaries under
profiles/<round>/: bench JSON lines, kernel-stat tables, and per-kernel HBM traffic from the PMC passes
(FETCH_SIZE is doubled on gfx950, both counters are in KB -- MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", rnd), os.path.join(root, "profiles", rnd)
os.makedirs(os.path.join(dst, "pmc"), exist_ok=True)

for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "*_probe.txt")) + [os.path.join(src, "gpu_tests.txt"), os.path.join(src, "smoke.txt"), os.path.join(src, "sigma_ladder.txt")]:
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, "final_" + os.path.basename(f)))
for d in glob.glob(os.path.join(src, "prof_*")):
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, "final_" + os.path.basename(d)[5:] + "_kernel_stats.csv"))


def short(name):
    # (kernels of an anonymous namespace -- sqd_lists.hip -- carry "(anonymous namespace)::" in front of their name)
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].replace("void ", "").strip()


# ---- sigma kernels of a kernel trace with the EARLY-EXIT launches split out: the host keeps one sigma build enqueued
# ahead of the Davidson round it has seen finish, so every solve ends with one or more launches that find the stop flag
# raised and return at once (< 1.5 us); averaged in, they flatter the kernel
for d in glob.glob(os.path.join(src, "prof_*")):
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        real, early, alld = defaultdict(list), defaultdict(int), defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = short(r["Kernel_Name"])
            if "k_sigma" not in name and "k_same_spin" not in name and "k_lists" not in name and "k_alpha_rows" not in name and "k_spmm" not in name and "k_opp" not in name:
                continue
            alld[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for name, ds in alld.items():
            # an early-exit launch still pays for dispatching its grid (0.8 us for 400 workgroups, ~4.5 us for the
            # thousands of a work-item or batched launch): cut at a third of the kernel's median, at least 1.5 us
            med = sorted(ds)[len(ds) // 2]
            cut = max(1.5, med / 3.0)
            for dur in ds:
                if dur < cut:
                    early[name] += 1
                else:
                    real[name].append(dur)
        if real:
            out = {k: {"launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v),
                       "early_exit_launches_excluded": early.get(k, 0)} for k, v in real.items()}
            json.dump(out, open(os.path.join(dst, "final_" + os.path.basename(d)[5:] + "_sigma_launches.json"), "w"), indent=1)


for wl in ("uniform317", "hf317", "big", "batch_uniform8", "batch_hf16", "hf1000", "hf3000", "hf5000"):
    out = {}
    for counter, sub in (("FETCH_SIZE", "pmc_fetch_"), ("WRITE_SIZE", "pmc_write_")):
        acc = defaultdict(list)
        for f in glob.glob(os.path.join(src, sub + wl, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        out[counter] = {k: {"dispatches": len(v), "avg_KB": sum(v) / len(v)} for k, v in acc.items()}
    if out["FETCH_SIZE"]:
        hbm = {}
        for k, v in out["FETCH_SIZE"].items():
            w = out["WRITE_SIZE"].get(k, {"avg_KB": 0.0})
            hbm[k] = {"dispatches": v["dispatches"], "hbm_bytes_per_launch": (2.0 * v["avg_KB"] + w["avg_KB"]) * 1024.0}
        out["HBM_BYTES"] = hbm
        if wl == "big":  # the list path's sigma is three launches: their sum is what one sigma moves
            parts = {k: v["hbm_bytes_per_launch"] for k, v in hbm.items()
                     if k.split("::")[-1].startswith(("k_lists_t4", "k_sigma_lists", "k_alpha_rows")) and "tab" not in k}
            if parts:
                out["LIST_PATH_SIGMA"] = {"kernels": parts, "hbm_bytes_per_sigma": sum(parts.values()),
                                          "note": "2 x FETCH_SIZE + WRITE_SIZE per launch (KB counters; the gfx950 "
                                                  "correction for wide coalesced reads), summed over the launches of one sigma"}
        if wl in ("hf1000", "hf3000"):  # connected sets: one sigma = two transpositions + the sparse product + the whole-row kernel
            parts = {k: v["hbm_bytes_per_launch"] * (2.0 if "transpose" in k else 1.0) for k, v in hbm.items()
                     if k.split("::")[-1].startswith(("k_spmm_grouped", "k_spmm_transpose", "k_opp_rows", "k_opp_reduce"))}
            if parts:
                out["CONNECTED_SIGMA"] = {"kernels": parts, "hbm_bytes_per_sigma": sum(parts.values()),
                                          "note": "2 x FETCH_SIZE + WRITE_SIZE per launch, summed over the launches of one sigma "
                                                  "(k_spmm_transpose runs twice per sigma: counted twice)"}
        name = wl if wl != "big" else "uniform10000"
        json.dump(out, open(os.path.join(dst, "pmc", f"final_{name}_pmc_summary.json"), "w"), indent=1)
        top = sorted(hbm.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:6]
        print(wl, [(k, round(v["hbm_bytes_per_launch"] / 1e6, 3), v["dispatches"]) for k, v in top])
# ---- matrix-core utilisation of the dense same-spin product (batched HF-centred solves): MFMA-pipe busy cycles over the
# kernel's busy cycles, per kernel, straight from the counters
for sub in ("pmc_mfma_batch_hf16", "pmc_mfma2_batch_hf16"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        out = {k: {c: {"dispatches": len(v), "avg": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()}
        # kernel durations of the same command (kernel trace of the batched HF-centred run), for the normalisation below
        dur_ns = {}
        for f in glob.glob(os.path.join(src, "prof_batch_hf16", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur_ns[short(r["Name"])] = float(r["AverageNs"])
        for k, d in out.items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]["avg"] > 0:
                d["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / d["SQ_BUSY_CYCLES"]["avg"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d and k in dur_ns:
                # fraction of the launch during which a SIMD's matrix pipe is busy: the counter is summed over the
                # chip's 1024 SIMDs (256 CUs x 4), in shader-clock cycles (2.4 GHz under load)
                cyc = dur_ns[k] * 1e-9 * 2.4e9
                d["avg_launch_ns_kernel_trace"] = dur_ns[k]
                d["mfma_pipe_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / (1024.0 * cyc)
                d["mfma_pipe_busy_frac_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x avg launch ns x 2.4 GHz)"
            if "SQ_INSTS_VALU_MFMA_MOPS_F64" in d and k in dur_ns:
                # MOPS_F64 counts 512 flops each (MI355X_MICROARCH / rocprof counter definition): achieved TFLOP/s
                d["tflops_from_mops_counter"] = d["SQ_INSTS_VALU_MFMA_MOPS_F64"]["avg"] * 512.0 / (dur_ns[k] * 1e-9) / 1e12
        json.dump(out, open(os.path.join(dst, "pmc", f"final_{sub[4:]}_summary.json"), "w"), indent=1)
        print(sub, {k: {c: (round(v["avg"], 1) if isinstance(v, dict) else (round(v, 4) if isinstance(v, float) else v)) for c, v in d.items()} for k, d in out.items() if "mfma" in k})
f = os.path.join(src, "pmc_lists", "summary.txt")
if os.path.exists(f):  # per-kernel counters of the list-pass sigma at 10^4 x 10^4 (profiles/probes/_pmc_lists.sh)
    shutil.copy(f, os.path.join(dst, "pmc", "final_lists_uniform10000_counters.txt"))
print("written to", dst)
