#!/bin/bash
# Collects the measurements that DESIGN.md section 8 and profiles/rNN/ quote.  Run on the GPU box:
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r01'
# Everything lands under gpurun_out/<round>/; profiles/summarize.py then writes the tracked summaries.
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -q > $OUT/gpu_tests.txt 2>&1
python bench.py > $OUT/bench_uniform317.json 2> $OUT/bench_uniform317.err
python bench.py --strings hf --skip-cpu > $OUT/bench_hf317.json 2>/dev/null
python bench.py --strings hf --spin-sq 0 --skip-cpu > $OUT/bench_hf317_spin0.json 2>/dev/null
python bench.py --skip-cpu --extra > $OUT/bench_extra_ladder.json 2>/dev/null
python bench.py --norb 40 --nelec 15 --na 707 --nb 707 --skip-cpu > $OUT/bench_fes_uniform707.json 2>/dev/null
python bench.py --norb 40 --nelec 15 --na 707 --nb 707 --strings hf --skip-cpu --steps 5 --warmup 1 > $OUT/bench_fes_hf707.json 2>/dev/null
python bench_pauli.py > $OUT/bench_pauli.json 2>/dev/null
python profiles/probes/_phase_probe.py > $OUT/phase_probe.txt 2>&1
python profiles/probes/_rdm_probe.py > $OUT/rdm_probe.txt 2>&1
python profiles/probes/_concurrency_probe.py > $OUT/concurrency_probe.txt 2>&1
python profiles/probes/_loop_probe.py > $OUT/loop_probe.txt 2>&1
CONC=6 python profiles/probes/_loop_probe.py >> $OUT/loop_probe.txt 2>&1
python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
cd /tmp && export TMPDIR=/tmp
# kernel traces of the SAME commands as the bench lines
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_uniform317 -o p -- python $ROOT/bench.py --skip-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hf317 -o p -- python $ROOT/bench.py --strings hf --skip-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fes_hf707 -o p -- python $ROOT/bench.py --norb 40 --nelec 15 --na 707 --nb 707 --strings hf --skip-cpu --steps 5 --warmup 1 > /dev/null 2>&1
# HBM traffic counters: separate passes, nothing else enabled (MI355X_MICROARCH.md, HBM / rocprofv3 section)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_uniform317 -o p -- python $ROOT/bench.py --skip-cpu --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_uniform317 -o p -- python $ROOT/bench.py --skip-cpu --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_hf317 -o p -- python $ROOT/bench.py --strings hf --skip-cpu --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_hf317 -o p -- python $ROOT/bench.py --strings hf --skip-cpu --steps 5 --warmup 1 > /dev/null 2>&1
# keep only what travels back comfortably (the merge limit is 64 MiB)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT
