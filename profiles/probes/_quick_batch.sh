#!/bin/bash
# Quick look while tuning the batched solve (GPU box): kernel averages of 16 HF-centred 317^2 subspaces solved as one batch.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/quickb
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CASE=${CASE:-hf16} REPS=${REPS:-5} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $ROOT/profiles/probes/_batch_trace.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('$OUT/prof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']:>6s} %")
PY
