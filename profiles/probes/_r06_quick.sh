#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
for cfg in "N=5000" "N=3000 SQD_OPP_SRC=1"; do echo "--- $cfg"; env $cfg timeout 300 python profiles/probes/_oppsrc_clock.py 2>&1 | eval $F; done
SIZES="${SIZES:-4000 5000 7000}" MODES=default CHECK=1 DAV=0 REPS=6 python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | cut -c1-120
SQD_OPP_SRC=1 SIZES="2000 3000" MODES=default CHECK=1 DAV=0 REPS=6 python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | cut -c1-120
