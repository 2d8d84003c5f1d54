#!/bin/bash
# rocprofv3 counter passes of the r02 build (GPU box, repo root).  Counters in their own runs, separate passes for
# FETCH_SIZE and WRITE_SIZE (TCC slots) and for the SQ set; only --pmc (no trace domains).  $1 = precision (f32 | bf16x3 | bf16).
export TMPDIR=/tmp
P=${1:-f32}
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d gpurun_out/pmc_${P}_fetch -o r02 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-settle --precision $P > /dev/null 2> gpurun_out/pmc_${P}_fetch.err
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_${P}_write -o r02 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-settle --precision $P > /dev/null 2> gpurun_out/pmc_${P}_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d gpurun_out/pmc_${P}_sq -o r02 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-settle --precision $P > /dev/null 2> gpurun_out/pmc_${P}_sq.err
ls gpurun_out/pmc_${P}_*/
