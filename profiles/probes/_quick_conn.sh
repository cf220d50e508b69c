cd $GRAFT_REPO_ROOT
SIZES="1000 2000 3000" MODES="spmm1" DAV=1 CHECK=1 python profiles/probes/_connected_probe.py 2>&1 | grep "^hf" | sed 's/ B_sigma.*links=[^ ]* *//' | cut -c1-260
