#!/bin/bash
# Connected (HF-centred) subspaces at D = 1e6 .. 1e7: probe lines + kernel traces of sigma-only runs (GPU box).
# usage: TAG=r05/conn0 SIZES="1000 3000" bash profiles/probes/_connected.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r05/conn}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
SIZES="${SIZES:-1000 2000 3000}" python profiles/probes/_connected_probe.py 2>&1 | eval $F | tee $OUT/connected_probe.txt
cd /tmp && export TMPDIR=/tmp
for n in ${TRACE_SIZES:-1000 3000}; do
  SIZES=$n MODES="${TRACE_MODES:-default}" CHECK=0 DAV=${TRACE_DAV:-1} REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hf$n -o p -- python $ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$OUT/prof_hf$n/**/*kernel_stats.csv', recursive=True)
if f:
    print('--- kernel stats hf $n')
    for r in list(csv.DictReader(open(f[0])))[:14]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Percentage']:>6s} %")
PY
done 2>&1 | tee $OUT/connected_kernel_stats.txt
find $OUT -name "*kernel_trace.csv" -size +12M -delete
find $OUT -name "*.db" -delete
