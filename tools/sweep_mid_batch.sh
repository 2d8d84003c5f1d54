#!/bin/bash
# mid-size batches (between the N-split small-batch kernel and the throughput shape): default workgroup shape against the forced
# throughput shape (VPC_TILE=128: for plain bf16 that is the whole-step kernel), per precision
mkdir -p gpurun_out/r03
out=gpurun_out/r03/${1:-mid_batch}.txt; rm -f $out
for prec in bf16 f32; do
  for B in 6144 8192 12288 16384 24576; do
    for env in "" "VPC_TILE=128"; do
      ms=$(env $env timeout -k 10 120 python bench.py --batch $B --precision $prec --no-cpu-baseline --no-extra-configs --steps 200 --warmup 30 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
      echo "$prec B=$B ${env:-default} $ms ms" | tee -a $out
    done
  done
done
