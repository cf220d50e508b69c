#!/bin/bash
# GPU probe: where the cycles of the sparse-product same-spin kernel go (SQ / TCP / TCC counters, separate passes)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r05/pmc_spmm}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*" | sort -u > $OUT/counters_available.txt
N=${N:-3000}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  SIZES=$N MODES=${MODE:-spmm1} CHECK=0 DAV=0 REPS=4 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -o p -- python $ROOT/profiles/probes/_connected_probe.py > /dev/null 2> $OUT/pass$i.err
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob('$OUT/pass*/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'spmm' not in k and 'k_sigma' not in k and 'mfma' not in k and 'k_opp' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Counter_Name'], r['Dispatch_Id'])
        if (k, r['Dispatch_Id'], r['Counter_Name']) not in seen:
            seen.add((k, r['Dispatch_Id'], r['Counter_Name']))
    # dispatch counts per kernel per counter
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'spmm' not in k and 'k_sigma' not in k and 'mfma' not in k and 'k_opp' not in k: continue
        calls[(k, r['Counter_Name'])] += 1
for k in agg:
    print('==', k)
    for c, v in sorted(agg[k].items()):
        n = calls[(k, c)]
        print(f"   {c:36s} {v / max(n, 1):16.1f} per dispatch ({n} samples)")
PY
find $OUT -name "*.db" -delete
