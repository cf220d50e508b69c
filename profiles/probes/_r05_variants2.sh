cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "row_sharded_overlap" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | grep "Error\|assert\|passed\|failed" | cut -c1-600 | tail -8
run() { echo "== $*"; env "$@" SIZES="${SIZES:-1000 2000}" MODES="spmm1" DAV=${DAV:-0} CHECK=${CHECK:-0} python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-230; }
run SQD_OPP_T=512
run SQD_OPP_T=1024
SIZES="700 900" run SQD_OPP_T=512
SIZES="700 900" MODES="dense1" run X=1
