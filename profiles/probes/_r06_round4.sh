#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
probe() {  # label, sizes, env...
  local label=$1 sizes=$2; shift 2
  echo "=== $label"
  env "$@" SIZES="$sizes" MODES=default CHECK=1 DAV=${DAV:-0} REPS=10 timeout 600 python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-260
}
probe "default selection (k_opp_rows <= 3072; k_opp_src + side-stream product beyond)" "3000 4000 5000 7000"
probe "k_opp_src forced, product on the side stream" "1000 2000 3000" SQD_OPP_SRC=1
probe "k_opp_src forced, product in line" "1000 2000 3000 5000" SQD_OPP_SRC=1 SQD_SIGMA_OVERLAP=0
probe "k_opp_src forced, side stream, 1024 threads" "2000 3000" SQD_OPP_SRC=1 SQD_OPPS_T=1024
DAV=1 probe "whole solves, default selection" "3000 5000"
DAV=1 probe "whole solves, k_opp_src + side stream forced" "2000 3000" SQD_OPP_SRC=1
echo "=== GPU tests: long rows + connected"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "long_rows or connected" 2>&1 | eval $F | tail -4
cd /tmp
for n in 3000 5000; do
  SQD_OPP_SRC=1 SIZES=$n MODES=default CHECK=0 DAV=0 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hf$n -o p -- python $GRAFT_REPO_ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('/tmp/prof_hf$n/**/*kernel_stats.csv', recursive=True)
if f:
    print('--- kernel stats hf $n (sigma only, k_opp_src + side stream)')
    for r in list(csv.DictReader(open(f[0])))[:7]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Percentage']:>6s} %")
PY
done
