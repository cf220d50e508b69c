#!/bin/bash
# GPU probe: k_alpha_part (part of a wavefront per row, XCD-private L2-resident panels) against k_alpha_rows
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" python profiles/probes/_big_sigma_probe.py 2>&1 | grep sigma; }
for N in 10000 6000; do
  F=""; [ $N -lt 8500 ] && F="SQD_SIGMA_LISTS=1"
  for L in 64 32 16; do for P in 4 15; do run N=$N $F SQD_ALPHA_LPR=$L SQD_LISTS_PASSES=$P; done; done
done
