#!/bin/bash
# Build a tuning variant of libsqd_hip.so: profiles/probes/build_variant.sh NAME "<flags for sqd_lists.hip ...>" [file.hip ...]
# Objects of the unflagged sources are cached under profiles/probes/_build/obj; the listed files (default: sqd_lists.hip)
# are compiled with the flags.  Result: profiles/probes/_build/libsqd_hip_NAME.so (SQD_LIB=... for the probes).
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CS=$ROOT/qiskit-addon-sqd_amd/csrc
B=$ROOT/profiles/probes/_build
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-sqd_lists.hip}
mkdir -p $B/obj
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$ROOT/include -I$CS"
OBJS=""
for f in sqd_tables sqd_sigma sqd_lists sqd_spmm sqd_opp sqd_oppsrc sqd_davidson sqd_rdm sqd_pauli sqd_recover sqd_capi; do
  if echo " $FILES " | grep -q " $f.hip "; then
    $CC $FLAGS -c $CS/$f.hip -o $B/obj/${f}_$NAME.o; OBJS="$OBJS $B/obj/${f}_$NAME.o"
  else
    if [ ! -f $B/obj/$f.o ] || [ $CS/$f.hip -nt $B/obj/$f.o ] || [ -n "$(find $CS -name '*.h' -newer $B/obj/$f.o)" ]; then $CC -c $CS/$f.hip -o $B/obj/$f.o; fi
    OBJS="$OBJS $B/obj/$f.o"
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $B/libsqd_hip_$NAME.so
echo built $B/libsqd_hip_$NAME.so
