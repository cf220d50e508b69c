#!/bin/bash
# GPU probe: per-launch durations of the Davidson BLAS-1 kernels at D = N^2 (default 1e8) from a rocprofv3 kernel trace,
# with the bandwidth each launch reaches (bytes = what the kernel must move once at basis size m).  env N, SQD_LIB.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=$ROOT/gpurun_out/prof_bigdav_$$
rocprofv3 --kernel-trace --output-format csv -d $D -o p -- python $ROOT/profiles/probes/_big_davidson_probe.py 2>&1 | grep "^{" 
python - "$D" "${N:-10000}" <<'PY'
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
D = n * n
seq = {"k_residual_precond": [], "k_dots_s": [], "k_dots_eig": [], "k_orth_dev": []}
for r in rows:
    for k in seq:
        if k + "<" in r["Kernel_Name"]:
            seq[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in seq.items():
    live = [x for x in v if x > 50]  # (early-exit launches of the speculative round ahead take ~5 us)
    if not live:
        continue
    # the second solve of the probe repeats the first: basis size m = 1, 2, ... per live launch, restarting
    out, m = [], 0
    for x in live:
        m += 1
        if out and x < 0.6 * out[-1][1] and m > 2:
            m = 1
        nbytes = {"k_residual_precond": (2 * m + 2), "k_dots_s": (m + 1), "k_dots_eig": (m + 1), "k_orth_dev": (m + 3)}[k] * 8 * D
        out.append((m, x, nbytes / x / 1e6))
    print(k, " ".join(f"m={m}:{x:7.1f}us={bw:5.2f}TB/s" for m, x, bw in out))
PY
rm -rf $D
