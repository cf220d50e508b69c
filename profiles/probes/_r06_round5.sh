#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
echo "=== sharded probe (one native call per iteration on a group of one)"
python profiles/probes/_sharded_probe.py 2>&1 | eval $F
echo "=== dots / eigen: fused (k_dots_eig, 119 VGPRs since round 6) against split, whole solves"
for d in 200000 100000000; do
  echo "--- SQD_DOTS_SPLIT_D=$d"
  SQD_DOTS_SPLIT_D=$d SIZES="500 700 1000 2000 3000" MODES=default CHECK=0 DAV=1 REPS=5 python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-200
done
echo "=== new GPU tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "single_pass_r16 or row_shard" 2>&1 | eval $F | tail -4
