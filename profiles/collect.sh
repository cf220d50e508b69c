#!/bin/bash
# Collects the measurements that DESIGN.md section 8 and profiles/rNN/ quote.  Run on the GPU box:
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r04'
# Everything lands under gpurun_out/<round>/; profiles/summarize.py then writes the tracked summaries.
set -u
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
F='grep -v -e amdgpu.ids -e RCCL -e "HIP version" -e "ROCm version" -e Hostname -e Librccl -e socket.cpp'
python -m pytest tests -m gpu -q 2>&1 | eval $F | tail -5 > $OUT/gpu_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_uniform317.json 2> $OUT/bench_uniform317.err  # the driver's command line
python bench.py --strings hf --skip-cpu --skip-secondary > $OUT/bench_hf317.json 2>/dev/null
SQD_SIGMA_DENSE=0 python bench.py --strings hf --skip-cpu --skip-secondary > $OUT/bench_hf317_sparse_same_spin.json 2>/dev/null
python bench.py --strings hf --spin-sq 0 --skip-cpu --skip-secondary > $OUT/bench_hf317_spin0.json 2>/dev/null
python bench.py --skip-cpu --skip-secondary --extra > $OUT/bench_extra_ladder.json 2>/dev/null
python bench.py --norb 40 --nelec 15 --na 707 --nb 707 --skip-cpu --skip-secondary > $OUT/bench_fes_uniform707.json 2>/dev/null
python bench.py --norb 40 --nelec 15 --na 707 --nb 707 --strings hf --skip-cpu --skip-secondary --steps 5 --warmup 1 > $OUT/bench_fes_hf707.json 2>/dev/null
SQD_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --skip-cpu --skip-secondary > $OUT/bench_forced_dist_1gpu.json 2> $OUT/bench_forced_dist_1gpu.err
python bench_pauli.py > $OUT/bench_pauli.json 2>/dev/null
python profiles/probes/_batch_probe.py 2>&1 | eval $F > $OUT/batch_probe.txt
SQD_SIGMA_DENSE=0 python profiles/probes/_batch_probe.py 2>&1 | eval $F | grep hf > $OUT/batch_sparse_same_spin_probe.txt
python profiles/probes/_loop_probe2.py 2>&1 | eval $F > $OUT/loop_subspaces_probe.txt
MODE=4 python profiles/probes/_loop_probe2.py 2>&1 | eval $F > $OUT/loop_subspaces_4streams_probe.txt
python profiles/probes/_sharded_probe.py 2>&1 | eval $F > $OUT/sharded_probe.txt
python profiles/probes/_phase_probe2.py 2>&1 | eval $F > $OUT/phase_probe.txt
python profiles/probes/_jitter_probe.py 2>&1 | eval $F > $OUT/jitter_probe.txt
python profiles/probes/_big_sigma_probe.py 2>&1 | eval $F > $OUT/big_sigma_probe.txt
SQD_SIGMA_LISTS=0 python profiles/probes/_big_sigma_probe.py 2>&1 | eval $F >> $OUT/big_sigma_probe.txt
SIZES="10000 6000 4000" bash profiles/probes/_lists_passes.sh 2>&1 | eval $F > $OUT/lists_passes_probe.txt
[ -f profiles/probes/_build/libsqd_hip_clk.so ] && python profiles/probes/_lists_clock.py 2>&1 | eval $F > $OUT/lists_clock_probe.txt
# phase clocks of the Davidson BLAS-1 kernels (probe build of the library: hipcc ... -DSQD_PHASE_CLOCK, see the probe's header)
[ -f profiles/probes/_build/libsqd_hip_clk.so ] && python profiles/probes/_phase_clock.py 2>&1 | eval $F > $OUT/phase_clock_probe.txt
[ -f profiles/probes/_build/libsqd_hip_clk.so ] && python profiles/probes/_sigma_clock.py 2>&1 | eval $F > $OUT/sigma_clock_probe.txt
[ -f profiles/probes/_build/libsqd_hip_clk.so ] && (for cfg in "N=5000" "N=3000 SQD_OPP_SRC=1"; do echo "--- $cfg"; env $cfg python profiles/probes/_oppsrc_clock.py 2>&1 | eval $F; done) > $OUT/oppsrc_clock_probe.txt
[ -x profiles/probes/anyorder/anyorder_probe ] && ./profiles/probes/anyorder/anyorder_probe > $OUT/anyorder_probe.txt 2>&1
python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
# CONNECTED subspaces at D = 1e6 .. 9e6 (round 5): every same-spin formulation, kernel traces of the default one
TAG=$R/conn SIZES="700 1000 2000 3000 4000 5000 7000" MODES="default dense1 dense0" TRACE_SIZES="1000 3000 5000" bash profiles/probes/_connected.sh > /dev/null 2>&1
cp $OUT/conn/connected_probe.txt $OUT/connected_probe.txt; cp $OUT/conn/connected_kernel_stats.txt $OUT/connected_kernel_stats_probe.txt
for n in 1000 3000 5000; do mkdir -p $OUT/prof_hf$n; cp -r $OUT/conn/prof_hf$n/* $OUT/prof_hf$n/ 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
# kernel traces of the SAME commands as the bench lines, and of the batched solves
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_uniform317 -o p -- python $ROOT/bench.py --skip-cpu --skip-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hf317 -o p -- python $ROOT/bench.py --strings hf --skip-cpu --skip-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fes_hf707 -o p -- python $ROOT/bench.py --norb 40 --nelec 15 --na 707 --nb 707 --strings hf --skip-cpu --skip-secondary --steps 5 --warmup 1 > /dev/null 2>&1
CASE=uniform8 REPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_batch_uniform8 -o p -- python $ROOT/profiles/probes/_batch_trace.py > /dev/null 2>&1
CASE=hf16 REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_batch_hf16 -o p -- python $ROOT/profiles/probes/_batch_trace.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_loop -o p -- python $ROOT/profiles/probes/_loop_probe2.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_big_sigma -o p -- python $ROOT/profiles/probes/_big_sigma_probe.py > /dev/null 2>&1
# HBM traffic counters: separate passes, nothing else enabled (MI355X_MICROARCH.md, HBM / rocprofv3 section)
for wl in uniform317 hf317; do
  S=""; [ $wl = hf317 ] && S="--strings hf"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$wl -o p -- python $ROOT/bench.py $S --skip-cpu --skip-secondary --steps 5 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$wl -o p -- python $ROOT/bench.py $S --skip-cpu --skip-secondary --steps 5 --warmup 1 > /dev/null 2>&1
done
for c in uniform8 hf16; do
  CASE=$c REPS=3 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_batch_$c -o p -- python $ROOT/profiles/probes/_batch_trace.py > /dev/null 2>&1
  CASE=$c REPS=3 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_batch_$c -o p -- python $ROOT/profiles/probes/_batch_trace.py > /dev/null 2>&1
done
# matrix-core utilisation of the dense same-spin product: busy cycles of the MFMA pipe against the kernel's cycles
CASE=hf16 REPS=3 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma_batch_hf16 -o p -- python $ROOT/profiles/probes/_batch_trace.py > /dev/null 2>&1
CASE=hf16 REPS=3 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_mfma2_batch_hf16 -o p -- python $ROOT/profiles/probes/_batch_trace.py > /dev/null 2>&1
for n in 1000 3000 5000; do
  SIZES=$n MODES=default CHECK=0 DAV=0 REPS=4 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_hf$n -o p -- python $ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
  SIZES=$n MODES=default CHECK=0 DAV=0 REPS=4 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_hf$n -o p -- python $ROOT/profiles/probes/_connected_probe.py > /dev/null 2>&1
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_big -o p -- python $ROOT/profiles/probes/_big_sigma_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_big -o p -- python $ROOT/profiles/probes/_big_sigma_probe.py > /dev/null 2>&1
# keep only what travels back comfortably (the merge limit is 64 MiB)
TAG=$R/pmc_lists bash $ROOT/profiles/probes/_pmc_lists.sh > $OUT/pmc_lists_probe.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +12M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
