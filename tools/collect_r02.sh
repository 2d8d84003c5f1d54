#!/bin/bash
# Everything profiles/ quotes for the r02 build, collected on the GPU box from the repo root (about 6 minutes):
# kernel statistics (rocprofv3 --kernel-trace --stats) of bench.py in the three precisions and of the small-batch shapes,
# PMC passes (f32, bf16x3), the default and the driver-like bench lines.  Text summaries land in gpurun_out/r02_collect/.
export TMPDIR=/tmp
O=gpurun_out/r02_collect; mkdir -p $O
bash tools/profile_r02.sh > $O/profile.log 2>&1
for p in f32 bf16x3 bf16; do python3 tools/rocpd_stats.py gpurun_out/prof_$p 20 > $O/r02_kernel_stats_$p.csv; cp gpurun_out/r02_bench_under_rocprof_$p.json $O/; done
for B in 8192 64; do python3 tools/rocpd_stats.py gpurun_out/prof_b$B 20 > $O/r02_kernel_stats_b$B.csv; done
for p in f32 bf16x3; do
  bash tools/pmc_r02.sh $p > $O/pmc_$p.log 2>&1
  for x in fetch write sq; do python3 tools/rocpd_pmc.py gpurun_out/pmc_${p}_$x 3 > $O/summary_${p}_$x.txt; done
done
python3 bench.py > $O/r02_bench_default.json 2> $O/bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_bench_driverlike_w5_s20.json 2>> $O/bench_default.err
for p in bf16x3 bf16; do python3 bench.py --no-cpu-baseline --precision $p > $O/r02_bench_$p.json 2>> $O/bench_default.err; done
python3 tools/bench_prec_small.py > $O/r02_small_prec.jsonl 2>> $O/bench_default.err
python3 tools/bench_small.py --graph --steps 600 > $O/r02_small_batch_steps.jsonl 2>> $O/bench_default.err
rm -rf gpurun_out/prof_* gpurun_out/pmc_*   # the rocpd databases are large; the summaries are what is kept
ls -la $O
