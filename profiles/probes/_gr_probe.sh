#!/bin/bash
# GPU probe: rows per group of the sparse same-spin product (probe builds of the library with -DSQD_SPMM_GR=4 / 16)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" SIZES="${SIZES:-1000 2000 3000}" MODES="spmm1" DAV=0 CHECK=${CHECK:-1} python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-200; }
run GR=8
run SQD_LIB=profiles/probes/_build/libsqd_hip_gr4.so
run SQD_LIB=profiles/probes/_build/libsqd_hip_gr16.so
run SQD_LIB=profiles/probes/_build/libsqd_hip_gr4.so SQD_SPMM_GJ=4
