#!/bin/bash
# GPU probe: the sparse-product same-spin kernels against their depth hooks (HF-centred N x N, whole sigma in us)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" SIZES="${SIZES:-1000 3000}" MODES="spmm1" DAV=0 CHECK=${CHECK:-0} python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-200; }
CHECK=1 run SQD_SPMM_GX=16
run SQD_SPMM_GX=8
run SQD_SPMM_GX=4
run SQD_SPMM_GROUPED=0 SQD_SPMM_U=16
run SQD_SPMM_GROUPED=0 SQD_SPMM_U=8
