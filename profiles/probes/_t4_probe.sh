#!/bin/bash
# GPU probe: k_lists_t4 variants (LIBS: library builds under profiles/probes/_build; "" = the in-tree library)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for L in ${LIBS:-intree t4}; do
  for P in ${PASSES:-2 15}; do
    if [ "$L" != intree ]; then export SQD_LIB=profiles/probes/_build/libsqd_hip_$L.so; else unset SQD_LIB; fi
    N=${N:-10000} SQD_SIGMA_LISTS=1 SQD_LISTS_PASSES=$P python profiles/probes/_big_sigma_probe.py 2>&1 | grep sigma
  done
done
