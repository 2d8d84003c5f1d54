#!/bin/bash
# rocprofv3 counter passes (GPU box, repo root).  Counters in their own runs - separate passes for FETCH_SIZE and WRITE_SIZE
# (TCC slots) and for the SQ set; only --pmc, no trace domains.  $1 = precision (f32 | bf16x3 | bf16), $2 = output tag.
export TMPDIR=/tmp
P=${1:-bf16}
T=${2:-r03}
O=gpurun_out/${T}_pmc
mkdir -p $O
ARGS="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-settle --precision $P"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/${P}_fetch -o $T -- python3 $ARGS > /dev/null 2> $O/${P}_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/${P}_write -o $T -- python3 $ARGS > /dev/null 2> $O/${P}_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $O/${P}_sq -o $T -- python3 $ARGS > /dev/null 2> $O/${P}_sq.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA -d $O/${P}_sq2 -o $T -- python3 $ARGS > /dev/null 2> $O/${P}_sq2.err
for k in fetch write sq sq2; do python3 tools/rocpd_pmc.py $O/${P}_$k 3 > $O/summary_${P}_$k.txt 2>&1; done
cat $O/summary_${P}_*.txt | grep -A 12 "step_bf16\|dec8\|enc_" | head -120
