#!/bin/bash
# rocprofv3 kernel statistics of the r02 build (run on the GPU box from the repo root): writes gpurun_out/prof_*/ and the
# bench lines of the same runs.  Counters (--pmc) are collected by tools/pmc_r02.sh in separate passes.
export TMPDIR=/tmp
for p in f32 bf16x3 bf16; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$p -o r02 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --precision $p > gpurun_out/r02_bench_under_rocprof_$p.json 2> gpurun_out/prof_$p.err
done
for B in 8192 64; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b$B -o r02 -- python3 tools/bench_small.py --batches $B --steps 100 > gpurun_out/r02_small_under_rocprof_b$B.jsonl 2> gpurun_out/prof_b$B.err
done
find gpurun_out -name "*kernel_stats.csv" | head -20
