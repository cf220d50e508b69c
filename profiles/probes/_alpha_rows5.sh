#!/bin/bash
# GPU probe: the list pass with every load of its row loop unconditional (exact wait counts), variants, phase clocks
cd ${GRAFT_REPO_ROOT:-/root/repo}
B=profiles/probes/_build
run() { env "$@" python profiles/probes/_big_sigma_probe.py 2>&1 | grep sigma; }
for P in 15 8 4; do run N=10000 SQD_LISTS_PASSES=$P; done
for V in pf0 pf3 spread0; do for P in 15 8; do run N=10000 SQD_LIB=$B/libsqd_hip_$V.so SQD_LISTS_PASSES=$P; done; done
for D in 4 2; do run N=10000 SQD_LISTS_PASSES=8 SQD_LISTS_DBG=$D; done
SQD_LIB=$B/libsqd_hip_clk.so python profiles/probes/_lists_clock.py 2>&1 | tail -2
