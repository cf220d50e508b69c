#!/bin/bash
# The round-5 subset of profiles/collect.sh: GPU tests, the driver's bench line (its secondaries carry the connected
# subspaces), the connected probe with kernel traces, and the HBM counters of one sigma at HF-centred 1000^2 / 3000^2.
#   gpurun --timeout 2400 -- 'bash profiles/collect_connected.sh r05'
set -u
R=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
python -m pytest tests -m gpu -q 2>&1 | eval $F | tail -5 > $OUT/gpu_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_uniform317.json 2> $OUT/bench_uniform317.err
python bench.py --strings hf --skip-cpu --skip-secondary > $OUT/bench_hf317.json 2>/dev/null
TAG=$R/conn SIZES="700 1000 2000 3000" MODES="default dense1 dense0" TRACE_SIZES="1000 3000" bash profiles/probes/_connected.sh > /dev/null 2>&1
cp $OUT/conn/connected_probe.txt $OUT/connected_probe.txt; cp $OUT/conn/connected_kernel_stats.txt $OUT/connected_kernel_stats_probe.txt
for n in 1000 3000; do mkdir -p $OUT/prof_hf$n; cp -r $OUT/conn/prof_hf$n/* $OUT/prof_hf$n/ 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
for n in 1000 3000; do
  SIZES=$n MODES=default CHECK=0 DAV=0 REPS=4 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_hf$n -o p -- python $ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
  SIZES=$n MODES=default CHECK=0 DAV=0 REPS=4 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_hf$n -o p -- python $ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
done
TAG=$R/pmc_spmm N=3000 MODE=default bash $ROOT/profiles/probes/_pmc_spmm.sh > $OUT/pmc_connected_hf3000_probe.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +12M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
