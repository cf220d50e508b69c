python profiles/probes/_concurrency_probe2.py 2>&1 | grep -v amdgpu
python profiles/probes/_jitter_probe.py 2>&1 | grep -v amdgpu | tail -6
python -m pytest tests -x -q -m gpu -k "config3 or concurrent or headline or full_parity" 2>&1 | grep -E "passed|failed|error" | tail -2
