#!/bin/bash
# GPU probe: columns per lane of k_spmm_grouped (scalar-cache stream per multiply-add halves / quarters)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" SIZES="${SIZES:-1000 2000 3000}" MODES="spmm1" DAV=0 CHECK=${CHECK:-0} python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-200; }
CHECK=1 run SQD_SPMM_GJ=2
run SQD_SPMM_GJ=1
run SQD_SPMM_GJ=4
run SQD_SPMM_GJ=2 SQD_SPMM_XCD=0
