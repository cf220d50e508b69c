P=profiles/probes/_big_sigma_probe.py
for j in 0 2 3; do SQD_ROWS_J=$j N=10000 python $P 2>&1 | grep -v amdgpu; SQD_ROWS_J=$j N=8000 python $P 2>&1 | grep -v amdgpu; done
python profiles/probes/_phase_probe2.py 2>&1 | grep -v amdgpu.ids
