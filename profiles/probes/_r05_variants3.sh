cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" SIZES="${SIZES:-1000 2000 3000}" MODES="${MODES:-spmm1}" DAV=${DAV:-1} CHECK=${CHECK:-0} python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-230; }
CHECK=1 run X=1
SIZES=317 MODES=dense1 run SQD_DOTS_SPLIT_D=50000
SIZES=317 MODES=dense1 run X=1
cd /tmp && export TMPDIR=/tmp
SIZES=3000 MODES=spmm1 CHECK=0 DAV=0 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05/q4/prof -o p -- python $GRAFT_REPO_ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('$GRAFT_REPO_ROOT/gpurun_out/r05/q4/prof/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:4]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us")
PY
