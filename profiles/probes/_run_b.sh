python profiles/probes/_phase_probe2.py 2>&1 | grep -v amdgpu.ids
python profiles/probes/_jitter_probe.py 2>&1 | grep -v amdgpu.ids | tail -6
