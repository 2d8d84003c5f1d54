#!/bin/bash
# decoder-kernel ablation: time the fused decoder with pieces switched off (diagnostic build)
export VPC_LIB=$PWD/vae-posterior-consistency_amd/csrc/libvpc_hip_ablate.so
for dbg in 0 1 2 4 8 6 14; do
  VPC_DEBUG=$dbg python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$dbg', 'dec_ms=%.4f'%d['kernels_ms']['decoder_fused'], 'step_ms=%.4f'%d['ms_per_step'])"
done
