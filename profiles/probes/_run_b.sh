P=profiles/probes/_big_sigma_probe.py
for n in 2000 3000 4000 6000 8000 10000 14000; do N=$n python $P 2>&1 | grep -v amdgpu; done
python -m pytest tests -x -q -m gpu -k "rows or sharded" 2>&1 | grep -E "passed|failed|error" | tail -3
