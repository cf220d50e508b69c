#!/bin/bash
# GPU probe: the list-path sigma at uniform N x N, launch by launch (SQD_LISTS_PASSES: bit 0 compact matrix, 1 compact
# term, 2 alpha side by rows, 3 the list pass), panel sizes of k_alpha_rows, and the build variants
# (profiles/probes/build_variant.sh: L2 prefetch distance / request rotation of the list pass; requests in flight and
# wavefronts per SIMD of k_alpha_rows).
cd ${GRAFT_REPO_ROOT:-/root/repo}
B=profiles/probes/_build
run() { env "$@" python profiles/probes/_big_sigma_probe.py 2>&1 | grep sigma; }
for N in ${SIZES:-10000}; do
  echo "== whole sigma and launch by launch"
  for P in 15 1 2 4 8; do run N=$N SQD_LISTS_PASSES=$P; done
  echo "== panel size (alpha side alone)"
  for MB in 8 16 32 96; do run N=$N SQD_LISTS_PASSES=4 SQD_ALPHA_PANEL_MB=$MB; done
  echo "== build variants: whole sigma, alpha side, list pass"
  for V in ${VARIANTS:-l2pf2 l2pf3 rotpf2 w7 k6 k12w5 k4}; do
    [ -f $B/libsqd_hip_$V.so ] || continue
    for P in 15 4 8; do run N=$N SQD_LIB=$B/libsqd_hip_$V.so SQD_LISTS_PASSES=$P; done
  done
  echo "== list pass without the next row's staging (dbg 4) / without gathers (dbg 2), default build and l2pf2"
  for D in 4 2; do run N=$N SQD_LISTS_PASSES=8 SQD_LISTS_DBG=$D; run N=$N SQD_LIB=$B/libsqd_hip_l2pf2.so SQD_LISTS_PASSES=8 SQD_LISTS_DBG=$D; done
done
