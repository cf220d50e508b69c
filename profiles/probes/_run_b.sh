P=profiles/probes/_big_sigma_probe.py
N=6000 python $P 2>&1 | grep -v amdgpu
SQD_SIGMA_ROWS=2 N=6000 python $P 2>&1 | grep -v amdgpu
N=5000 python $P 2>&1 | grep -v amdgpu
SQD_SIGMA_ROWS=0 N=5000 python $P 2>&1 | grep -v amdgpu
N=3000 python $P 2>&1 | grep -v amdgpu
SQD_SIGMA_ROWS=4 N=3000 python $P 2>&1 | grep -v amdgpu
SQD_DBG_NT=1 N=10000 python $P 2>&1 | grep -v amdgpu
SQD_DBG_NT=1 N=6000 python $P 2>&1 | grep -v amdgpu
SQD_DBG_NT=1 N=4000 python $P 2>&1 | grep -v amdgpu
