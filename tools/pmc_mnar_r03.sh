#!/bin/bash
# rocprofv3 counter passes of the MNAR step (config 3, plain bf16, B = $1): the layer-fused decoder kernel and - $2 = gemm - the
# GEMM chain it replaces (VPC_NMDEC=0).  Counters in their own runs (FETCH_SIZE, WRITE_SIZE, one SQ set); only --pmc.
export TMPDIR=/tmp
B=${1:-65536}
F=${2:-fused}
[ "$F" = "gemm" ] && export VPC_NMDEC=0
O=gpurun_out/r03_mnar_pmc
mkdir -p $O
ARGS="tools/bench_mnar.py --batch $B --precision bf16 --no-cpu --steps 6 --warmup 3"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/${F}_b${B}_fetch -o mnar -- python3 $ARGS > /dev/null 2> $O/${F}_b${B}_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/${F}_b${B}_write -o mnar -- python3 $ARGS > /dev/null 2> $O/${F}_b${B}_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $O/${F}_b${B}_sq -o mnar -- python3 $ARGS > /dev/null 2> $O/${F}_b${B}_sq.err
for k in fetch write sq; do python3 tools/rocpd_pmc.py $O/${F}_b${B}_$k 3 > $O/summary_${F}_b${B}_$k.txt 2>&1; done
head -40 $O/summary_${F}_b${B}_fetch.txt; head -30 $O/summary_${F}_b${B}_write.txt
