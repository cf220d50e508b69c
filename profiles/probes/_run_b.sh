P=profiles/probes/_big_sigma_probe.py
for n in 1000 1500 2000 3000 4000 6000 8000 10000 14000; do
  SQD_SIGMA_ROWS=0 N=$n python $P 2>&1 | grep -v amdgpu
  SQD_SIGMA_ROWS=8 N=$n python $P 2>&1 | grep -v amdgpu
done
SQD_SIGMA_ROWS=4 N=1500 python $P 2>&1 | grep -v amdgpu
SQD_SIGMA_ROWS=4 N=3000 python $P 2>&1 | grep -v amdgpu
SQD_SIGMA_ROWS=2 N=6000 python $P 2>&1 | grep -v amdgpu
