#!/bin/bash
# Collects the measurements that DESIGN.md section 8 and profiles/rNN/ quote.  Run on the GPU box:
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r02'
# Everything lands under gpurun_out/<round>/; profiles/summarize.py then writes the tracked summaries.
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_uniform317.json 2> $OUT/bench_uniform317.err  # the driver's command line
python bench.py --strings hf --skip-cpu --skip-secondary > $OUT/bench_hf317.json 2>/dev/null
python bench.py --strings hf --spin-sq 0 --skip-cpu --skip-secondary > $OUT/bench_hf317_spin0.json 2>/dev/null
python bench.py --skip-cpu --skip-secondary --extra > $OUT/bench_extra_ladder.json 2>/dev/null
python bench.py --norb 40 --nelec 15 --na 707 --nb 707 --skip-cpu --skip-secondary > $OUT/bench_fes_uniform707.json 2>/dev/null
python bench.py --norb 40 --nelec 15 --na 707 --nb 707 --strings hf --skip-cpu --skip-secondary --steps 5 --warmup 1 > $OUT/bench_fes_hf707.json 2>/dev/null
SQD_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --skip-cpu --skip-secondary > $OUT/bench_forced_dist_1gpu.json 2> $OUT/bench_forced_dist_1gpu.err
python bench_pauli.py > $OUT/bench_pauli.json 2>/dev/null
python profiles/probes/_phase_probe2.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_probe.txt
python profiles/probes/_jitter_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/jitter_probe.txt
python profiles/probes/_concurrency_probe2.py 2>&1 | grep -v amdgpu.ids > $OUT/concurrency_probe.txt
python profiles/probes/_loop_probe.py > $OUT/loop_probe.txt 2>&1
python profiles/probes/_big_sigma_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/big_sigma_probe.txt
# sigma of uniform n x n sets with the work-item kernel (SQD_SIGMA_ROWS=0) and with the default selection
: > $OUT/sigma_ladder.txt
for n in 1000 2000 3000 4000 6000 8000 10000 14000; do
  SQD_SIGMA_ROWS=0 N=$n python profiles/probes/_big_sigma_probe.py 2>&1 | grep -v amdgpu.ids >> $OUT/sigma_ladder.txt
  N=$n python profiles/probes/_big_sigma_probe.py 2>&1 | grep -v amdgpu.ids >> $OUT/sigma_ladder.txt
done
python profiles/probes/_eig_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/eig_probe.txt
python profiles/probes/_bench_gap_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_gap_probe.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29551 profiles/probes/_exchange_probe2.py 2>&1 | grep -v -e amdgpu.ids -e socket.cpp > $OUT/exchange_probe.txt
python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
cd /tmp && export TMPDIR=/tmp
# kernel traces of the SAME commands as the bench lines
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_uniform317 -o p -- python $ROOT/bench.py --skip-cpu --skip-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hf317 -o p -- python $ROOT/bench.py --strings hf --skip-cpu --skip-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fes_hf707 -o p -- python $ROOT/bench.py --norb 40 --nelec 15 --na 707 --nb 707 --strings hf --skip-cpu --skip-secondary --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_big_sigma -o p -- python $ROOT/profiles/probes/_big_sigma_probe.py > /dev/null 2>&1
# HBM traffic counters: separate passes, nothing else enabled (MI355X_MICROARCH.md, HBM / rocprofv3 section)
for wl in uniform317 hf317; do
  S=""; [ $wl = hf317 ] && S="--strings hf"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$wl -o p -- python $ROOT/bench.py $S --skip-cpu --skip-secondary --steps 5 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$wl -o p -- python $ROOT/bench.py $S --skip-cpu --skip-secondary --steps 5 --warmup 1 > /dev/null 2>&1
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_big -o p -- python $ROOT/profiles/probes/_big_sigma_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_big -o p -- python $ROOT/profiles/probes/_big_sigma_probe.py > /dev/null 2>&1
# keep only what travels back comfortably (the merge limit is 64 MiB)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
ls -la $OUT
