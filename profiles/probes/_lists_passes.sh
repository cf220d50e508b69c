#!/bin/bash
# GPU probe: the list-path sigma (sqd_lists.hip) at uniform N x N -- whole sigma, launch by launch (SQD_LISTS_PASSES: bit 1
# single x single term, 2 alpha side by rows, 3 the list pass; bit 0 unused), and round 3's k_sigma_rows on the same input.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for N in ${SIZES:-10000}; do
  for P in 15 2 4 8; do N=$N SQD_SIGMA_LISTS=1 SQD_LISTS_PASSES=$P python profiles/probes/_big_sigma_probe.py 2>&1 | grep sigma; done
  N=$N SQD_SIGMA_LISTS=0 python profiles/probes/_big_sigma_probe.py 2>&1 | grep sigma
done
