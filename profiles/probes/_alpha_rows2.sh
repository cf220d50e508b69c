#!/bin/bash
# GPU probe: defaults of the list-path sigma after the restructuring; spread-requests variant; the list pass's phase clocks;
# k_alpha_rows at sizes where a panel of 128 columns is L2-resident (is the pass bound by the Infinity Cache?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
B=profiles/probes/_build
run() { env "$@" python profiles/probes/_big_sigma_probe.py 2>&1 | grep sigma; }
echo "== defaults, N = 10000"
for P in 15 4 8; do run N=10000 SQD_LISTS_PASSES=$P; done
echo "== spread / no L2 prefetch"
for V in spread pf0; do for P in 15 8; do run N=10000 SQD_LIB=$B/libsqd_hip_$V.so SQD_LISTS_PASSES=$P; done; done
echo "== phase clocks of the list pass (default, spread)"
SQD_LIB=$B/libsqd_hip_clk.so python profiles/probes/_lists_clock.py 2>&1 | tail -2
SQD_LIB=$B/libsqd_hip_clkspread.so python profiles/probes/_lists_clock.py 2>&1 | tail -2
echo "== alpha side alone at smaller sizes (forced list path): bytes = 12 x 8 N^2"
for N in 2000 3000 4000 6000 8000; do run N=$N SQD_SIGMA_LISTS=1 SQD_LISTS_PASSES=4; done
echo "== whole sigma at smaller sizes: list path forced | default selection"
for N in 4000 6000 8000; do run N=$N SQD_SIGMA_LISTS=1; run N=$N; done
