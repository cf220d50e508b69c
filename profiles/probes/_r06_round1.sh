#!/bin/bash
# Round 6, first full pass on the GPU: the -m gpu suite, the bench line, the long-row probe.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
python -m pytest tests -m gpu -q -x 2>&1 | eval $F | tail -8
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench1.json 2> gpurun_out/r06_bench1.err; tail -c 1500 gpurun_out/r06_bench1.json
probe() {  # label, sizes, env...
  local label=$1 sizes=$2; shift 2
  echo "=== $label"
  env "$@" SIZES="$sizes" MODES=default CHECK=1 DAV=1 REPS=5 timeout 900 python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-260
}
probe "default selection" "1000 3000 4000 5000 7000"
