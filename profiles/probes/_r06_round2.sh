#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
echo "=== k_opp_src phase clocks"
for cfg in "N=3000 SQD_OPP_SRC=1" "N=3000 SQD_OPP_SRC=1 SQD_OPPS_T=1024" "N=5000" "N=1000 SQD_OPP_SRC=1"; do
  echo "--- $cfg"; env $cfg timeout 300 python profiles/probes/_oppsrc_clock.py 2>&1 | eval $F
done
echo "=== GPU tests: long rows"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "long_rows" 2>&1 | eval $F | tail -4
echo "=== headline bench under rocprofv3 --kernel-trace (calibration of the roofline leg's launch duration)"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_head -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu --skip-secondary > /tmp/bench_prof.json 2>/tmp/bench_prof.err
python - <<'PY'
import csv, glob, json
f = glob.glob('/tmp/prof_head/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Percentage']:>6s} %")
t = glob.glob('/tmp/prof_head/**/*kernel_trace.csv', recursive=True)
d = [ (float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(t[0])) if 'k_sigma_direct' in r['Kernel_Name']]
live = [x for x in d if x > 1.5]
import statistics
print('k_sigma_direct launches', len(d), 'live (>1.5 us)', len(live), 'mean live', sum(live)/len(live), 'median live', statistics.median(live))
b = json.load(open('/tmp/bench_prof.json'))
print('bench line under the profiler: ms_per_step', b['ms_per_step'], 'median', b.get('ms_per_step_median'), 'roofline', {k: b['roofline'][k] for k in ('avg_launch_ms','event_bracket_ms','empty_bracket_ms','timed_launches','frac')})
PY
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu --skip-secondary > gpurun_out/r06_bench2.json 2>/dev/null
python - <<'PY'
import json
b = json.load(open('gpurun_out/r06_bench2.json'))
print('bench line without profiler: ms_per_step', b['ms_per_step'], 'median', b.get('ms_per_step_median'), 'roofline', {k: b['roofline'][k] for k in ('avg_launch_ms','event_bracket_ms','empty_bracket_ms','timed_launches','frac')}, b['roofline']['in_region_samples'])
PY
