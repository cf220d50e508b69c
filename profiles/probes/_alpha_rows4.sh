#!/bin/bash
# GPU probe: list pass with the single links' J term folded into the first round, conflict-aware link order (SQD_LISTS_ORDER),
# spread requests; LDS counters of both link orders; the list path against the default selection at smaller sizes
cd ${GRAFT_REPO_ROOT:-/root/repo}
B=profiles/probes/_build
run() { env "$@" python profiles/probes/_big_sigma_probe.py 2>&1 | grep sigma; }
for O in 1 0; do for P in 15 8; do run N=10000 SQD_LISTS_ORDER=$O SQD_LISTS_PASSES=$P; done; done
for P in 15 8; do run N=10000 SQD_LIB=$B/libsqd_hip_spread0.so SQD_LISTS_PASSES=$P; done
echo "== smaller sizes: list path forced | default selection"
for N in 1500 2000 3000 4000 5000 7000; do run N=$N SQD_SIGMA_LISTS=1; run N=$N; done
echo "== LDS counters, list pass alone"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for O in 1 0; do
  SQD_LISTS_ORDER=$O SQD_LISTS_PASSES=8 N=10000 timeout 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_o$O -o p -- python $R/profiles/probes/_big_sigma_probe.py > /tmp/pmc_o$O.log 2>&1
  python - <<PY
import csv, glob, collections
res = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/pmc_o$O/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_sigma_lists' in row['Kernel_Name']:
            k = row['Counter_Name']; res[k][0] += float(row['Counter_Value']); res[k][1] += 1
print('order=$O', {k: f'{v/max(n,1):.4e}' for k, (v, n) in sorted(res.items())})
PY
done
