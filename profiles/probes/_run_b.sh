P=profiles/probes/_big_sigma_probe.py
for n in 317 707 1000 2000; do
  GEN=hf N=$n python $P 2>&1 | grep -v amdgpu
  for r in 2 8; do GEN=hf SQD_SIGMA_ROWS=$r N=$n python $P 2>&1 | grep -v amdgpu; done
done
