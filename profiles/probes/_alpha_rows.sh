#!/bin/bash
# GPU probe: the alpha side of the list-path sigma at uniform N x N -- by rows on C (k_alpha_rows, panels of several sizes)
# against the list pass on C^T, and the list passes with the request rotation / L2 prefetch variants
# (profiles/probes/build_variant.sh rot|l2pf2|l2pf3|rotpf).
cd ${GRAFT_REPO_ROOT:-/root/repo}
B=profiles/probes/_build
run() { env "$@" python profiles/probes/_big_sigma_probe.py 2>&1 | grep sigma; }
for N in ${SIZES:-10000}; do
  echo "== by rows, whole sigma and pass by pass"
  for P in 15 1 2 4 8; do run N=$N SQD_LISTS_PASSES=$P; done
  echo "== by rows, panel size (alpha pass alone)"
  for MB in 16 32 64 128 192 100000; do run N=$N SQD_LISTS_PASSES=4 SQD_ALPHA_PANEL_MB=$MB; done
  echo "== alpha as a list pass"
  for P in 15 1 4 8; do run N=$N SQD_LISTS_ALPHA=0 SQD_LISTS_PASSES=$P; done
  echo "== list-pass variants (alpha list pass 4, beta pass 8)"
  for V in rot l2pf2 l2pf3 rotpf; do
    [ -f $B/libsqd_hip_$V.so ] || continue
    for P in 4 8; do run N=$N SQD_LIB=$B/libsqd_hip_$V.so SQD_LISTS_ALPHA=0 SQD_LISTS_PASSES=$P; done
  done
  echo "== list pass without the next row's staging (dbg 4) / without gathers (dbg 2)"
  for D in 4 2; do for P in 4 8; do run N=$N SQD_LISTS_ALPHA=0 SQD_LISTS_PASSES=$P SQD_LISTS_DBG=$D; done; done
done
