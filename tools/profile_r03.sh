#!/bin/bash
# rocprofv3 kernel statistics (GPU box, repo root): $1 = precision, $2 = tag.  Counters are collected by tools/pmc_r03.sh.
export TMPDIR=/tmp
P=${1:-f32}
T=${2:-r03}
mkdir -p gpurun_out/${T}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof/$P -o $T -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra-configs --precision $P > gpurun_out/${T}_prof/bench_under_rocprof_$P.json 2> gpurun_out/${T}_prof/$P.err
python3 tools/rocpd_stats.py gpurun_out/${T}_prof/$P 10 > gpurun_out/${T}_prof/kernel_stats_$P.csv
cat gpurun_out/${T}_prof/kernel_stats_$P.csv
