#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
echo "=== k_opp_src phase clocks (weights through LDS, packed column table)"
for cfg in "N=3000 SQD_OPP_SRC=1" "N=3000 SQD_OPP_SRC=1 SQD_OPPS_T=1024" "N=5000" "N=1000 SQD_OPP_SRC=1"; do
  echo "--- $cfg"; env $cfg timeout 300 python profiles/probes/_oppsrc_clock.py 2>&1 | eval $F
done
probe() {  # label, sizes, env...
  local label=$1 sizes=$2; shift 2
  echo "=== $label"
  env "$@" SIZES="$sizes" MODES=default CHECK=1 DAV=0 REPS=10 timeout 600 python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-230
}
probe "default selection" "1000 3000 4000 5000 7000"
probe "k_opp_src forced" "1000 2000 3000" SQD_OPP_SRC=1
probe "k_opp_src forced, 1024 threads" "2000 3000" SQD_OPP_SRC=1 SQD_OPPS_T=1024
probe "k_opp_src, E=16" "3000 5000" SQD_OPP_SRC=1 SQD_OPPS_E=16
probe "k_opp_src, E=64" "3000 5000" SQD_OPP_SRC=1 SQD_OPPS_E=64
echo "=== GPU tests: long rows + connected"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "long_rows or connected" 2>&1 | eval $F | tail -4
