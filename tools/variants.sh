#!/bin/bash
for lib in libvpc_hip.so libvpc_hip_ablate.so; do
  VPC_LIB=$PWD/vae-posterior-consistency_amd/csrc/$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['kernels_ms'], 'step_ms=%.4f'%d['ms_per_step'])"
done
