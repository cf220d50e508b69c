#!/bin/bash
# Round 6: the source-range opposite-spin kernel -- parity of the connected tests, then sigma time and per-kernel split for
# geometry variants.  usage (GPU box): bash profiles/probes/_r06_opp.sh > gpurun_out/r06_opp.txt 2>&1
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
echo "=== GPU parity: connected tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "connected or penalty or long_rows" 2>&1 | eval $F | tail -6
probe() {  # label, sizes, env...
  local label=$1 sizes=$2; shift 2
  echo "=== $label"
  env "$@" SIZES="$sizes" MODES=default CHECK=1 DAV=0 REPS=10 timeout 600 python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-230
}
probe "default (T=512 S=8 E=32)" "1000 2000 3000 5000"
probe "T=1024" "1000 3000 5000" SQD_OPP_T=1024
probe "T=256" "1000 3000" SQD_OPP_T=256
probe "S=4" "1000 3000" SQD_OPP_S=4
probe "E=16" "1000 3000 5000" SQD_OPP_E=16
probe "E=64" "1000 3000 5000" SQD_OPP_E=64
probe "E=8" "1000 3000" SQD_OPP_E=8
cd /tmp
for n in 1000 3000 5000; do
  SIZES=$n MODES=default CHECK=0 DAV=0 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hf$n -o p -- python $GRAFT_REPO_ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('/tmp/prof_hf$n/**/*kernel_stats.csv', recursive=True)
if f:
    print('--- kernel stats hf $n (sigma only, default geometry)')
    for r in list(csv.DictReader(open(f[0])))[:8]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Percentage']:>6s} %")
PY
done
