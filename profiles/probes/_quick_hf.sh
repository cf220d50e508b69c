#!/bin/bash
# Quick look while tuning the Davidson iteration (GPU box): HF-centred 317^2 bench line, its kernel averages, the 16-batch probe.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
python bench.py --strings hf --skip-cpu --skip-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('hf317 ms_per_step', d['ms_per_step'], 'roofline', d['roofline'].get('frac'), 'iters', d['config'].get('n_sigma', d['config']))" 2>&1 | cut -c1-400
[ "${BATCH:-1}" = 1 ] && python profiles/probes/_batch_probe.py 2>&1 | eval $F | grep -e "hf 16" -e "uniform 8"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $ROOT/bench.py --strings hf --skip-cpu --skip-secondary > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('$OUT/prof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']:>6s} %")
PY
