#!/bin/bash
# GPU probe: hardware counters of the list-pass sigma at uniform N x N, per kernel (separate --pmc passes)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG:-pmc_lists}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  N=${N:-10000} timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o p -- python $ROOT/profiles/probes/_big_sigma_probe.py > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get('GRAFT_REPO_ROOT','/root/repo') + '/gpurun_out/' + os.environ.get('TAG','pmc_lists')
res = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row['Kernel_Name']
        if 'k_sigma_lists' in kn or 'k_lists_' in kn or 'k_alpha_rows' in kn or 'k_sigma_rows' in kn:
            import re
            name = re.search(r'k_\w+(<[^>]*>)?', kn).group(0)
            k = (name, row['Counter_Name']); res[k][0] += float(row['Counter_Value']); res[k][1] += 1
with open(out + '/summary.txt', 'w') as fh:
    for k, (v, n) in sorted(res.items()):
        line = f"{k[0]:28s} {k[1]:32s} per launch {v/max(n,1):.4e} launches {n}"
        print(line); fh.write(line + '\n')
PY
rm -rf $OUT/p[0-9]*
