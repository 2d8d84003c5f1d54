#!/bin/bash
# rocprofv3 kernel statistics of the MNAR step (config 3, plain bf16): $1 = batch, $2 = tag
export TMPDIR=/tmp
B=${1:-128}
T=${2:-r03_mnar}
mkdir -p gpurun_out/${T}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof/b$B -o $T -- python3 tools/bench_mnar.py --batch $B --precision bf16 --no-cpu --steps 100 --warmup 20 > gpurun_out/${T}_prof/bench_b$B.json 2> gpurun_out/${T}_prof/b$B.err
python3 tools/rocpd_stats.py gpurun_out/${T}_prof/b$B 20 > gpurun_out/${T}_prof/kernel_stats_b$B.csv
cat gpurun_out/${T}_prof/kernel_stats_b$B.csv
