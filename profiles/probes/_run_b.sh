P=profiles/probes/_big_sigma_probe.py
for d in 0 3 7 11 15 31 63 51 35; do SQD_DBG=$d N=10000 python $P 2>&1 | grep -v amdgpu; done
