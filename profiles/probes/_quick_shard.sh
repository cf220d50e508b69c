#!/bin/bash
# Quick look while tuning the row-sharded solver (GPU box): kernel averages of profiles/probes/_sharded_probe.py
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/quicks
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $ROOT/profiles/probes/_sharded_probe.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('$OUT/prof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']:>6s} %")
PY
