cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "row_sharded_overlap" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30
run() { echo "== $*"; env "$@" SIZES="${SIZES:-1000 2000 3000}" MODES="spmm1" DAV=${DAV:-0} CHECK=${CHECK:-0} python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-230; }
CHECK=1 DAV=1 run SQD_SPMM_LDS=1
run SQD_SPMM_LDS=0
run SQD_SPMM_LDS=1 SQD_SPMM_GJ=1
run SQD_SPMM_LDS=1 SQD_OPP_E=64
SIZES=317 run SQD_DOTS_SPLIT_D=50000 DAV=1
SIZES=317 run DAV=1
cd /tmp && export TMPDIR=/tmp
for n in 1000 3000; do
SIZES=$n MODES=spmm1 CHECK=0 DAV=0 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05/q2/prof_hf$n -o p -- python $GRAFT_REPO_ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('$GRAFT_REPO_ROOT/gpurun_out/r05/q2/prof_hf$n/**/*kernel_stats.csv', recursive=True)
if f:
    print('--- kernel stats hf $n')
    for r in list(csv.DictReader(open(f[0])))[:5]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Percentage']:>6s} %")
PY
done
