#!/bin/bash
# GPU probe: whole-row opposite-spin kernel (sqd_opp.hip) on / off and its hooks at HF-centred N x N, then the kernel trace
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
run() { echo "== $*"; env "$@" SIZES="${SIZES:-1000 2000 3000}" MODES="spmm1" DAV=${DAV:-0} CHECK=${CHECK:-0} python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-260; }
CHECK=1 DAV=1 run SQD_SIGMA_OPP=1
run SQD_SIGMA_OPP=1 SQD_OPP_E=4
run SQD_SIGMA_OPP=1 SQD_OPP_E=8
run SQD_SIGMA_OPP=1 SQD_OPP_E=16
run SQD_SIGMA_OPP=1 SQD_OPP_E=32
run SQD_SIGMA_OPP=1 SQD_OPP_T=1024 SQD_OPP_E=8
run SQD_SIGMA_OPP=0
cd /tmp && export TMPDIR=/tmp
for n in 1000 3000; do
SIZES=$n MODES=spmm1 CHECK=0 DAV=0 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05/opp/prof_hf$n -o p -- python $ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('$ROOT/gpurun_out/r05/opp/prof_hf$n/**/*kernel_stats.csv', recursive=True)
if f:
    print('--- kernel stats hf $n')
    for r in list(csv.DictReader(open(f[0])))[:6]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Percentage']:>6s} %")
PY
done
