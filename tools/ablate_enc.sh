#!/bin/bash
export VPC_LIB=$PWD/vae-posterior-consistency_amd/csrc/libvpc_hip_ablate.so
for dbg in 0 1 2 4 6 7; do
  VPC_DEBUG_ENC=$dbg python bench.py --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('enc dbg=$dbg', 'enc_bwd_ms=%.4f'%d['kernels_ms']['encoder_bwd'])"
done
