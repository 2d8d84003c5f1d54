#!/bin/bash
# whole-step bf16 kernel: kernel time with pieces switched off (diagnostic build; results are wrong by design)
#   VPC_DEBUG 2 = no workgroup barriers, 8 = no touch prefetch, 16 = no tile-input loads (staged path)
python vae-posterior-consistency_amd/csrc/build.py --ablate > /dev/null 2>&1
export VPC_LIB=$PWD/vae-posterior-consistency_amd/csrc/libvpc_hip_ablate.so
for dbg in ${@:-0 2 16 18}; do
  VPC_DEBUG=$dbg python bench.py --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$dbg', 'step_fused_ms=%.4f'%d['kernels_ms']['step_fused'], 'step_ms=%.4f'%d['ms_per_step'])"
done
