#!/bin/bash
# rocprofv3 kernel statistics of the EDDI step (Reg_EDDI, d = 128, batch 64)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_eddi_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_eddi_prof/b64 -o eddi -- python3 tools/bench_eddi.py --batch 64 --steps 100 --warmup 20 --no-cpu > gpurun_out/r03_eddi_prof/bench_b64.json 2> gpurun_out/r03_eddi_prof/b64.err
python3 tools/rocpd_stats.py gpurun_out/r03_eddi_prof/b64 30 > gpurun_out/r03_eddi_prof/kernel_stats_b64.csv
cat gpurun_out/r03_eddi_prof/kernel_stats_b64.csv | cut -c1-150
