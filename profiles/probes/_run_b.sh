mkdir -p gpurun_out/r02b
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r02b/gputests.txt
python profiles/probes/_phase_probe2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02b/phase_probe.txt
python profiles/probes/_jitter_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02b/jitter_probe.txt
python profiles/probes/_bench_gap_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02b/bench_gap_probe.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu --skip-secondary > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
python bench.py --gpus 1 --steps 200 --warmup 5 --skip-cpu --skip-secondary > gpurun_out/r02b/bench200.json 2> gpurun_out/r02b/bench200.err
cat gpurun_out/r02b/gputests.txt gpurun_out/r02b/phase_probe.txt gpurun_out/r02b/jitter_probe.txt gpurun_out/r02b/bench_gap_probe.txt
python - <<'PY'
import json
for f in ('bench','bench200'):
    d=json.loads(open(f'gpurun_out/r02b/{f}.json').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], d.get('native_ms_per_step'), d['roofline']['avg_launch_ms'], d['roofline']['frac'])
PY
