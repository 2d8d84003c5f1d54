#!/bin/bash
# layer-fused MNAR decoder kernel: in-kernel phase stamps (diagnostic build, VPC_DEBUG 64) and timing without barriers (2)
mkdir -p gpurun_out/r03
python vae-posterior-consistency_amd/csrc/build.py --ablate > gpurun_out/r03/nmdec_ablate_build.log 2>&1 || { tail -5 gpurun_out/r03/nmdec_ablate_build.log; exit 1; }
export VPC_LIB=$PWD/vae-posterior-consistency_amd/csrc/libvpc_hip_ablate.so
B=${1:-65536}
for dbg in 64 2; do
  echo "== VPC_DEBUG=$dbg B=$B"
  VPC_DEBUG=$dbg timeout -k 10 200 python tools/bench_mnar.py --batch $B --precision bf16 --no-cpu --steps 3 --warmup 1 --timers 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-1500
done
