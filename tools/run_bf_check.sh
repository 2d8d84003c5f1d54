python -m pytest tests/test_bf16.py -m gpu -x -q > gpurun_out/t_bf.log 2>&1; tail -2 gpurun_out/t_bf.log
python bench.py --no-cpu-baseline --precision bf16x3 > gpurun_out/b_x3.json 2>gpurun_out/b_x3.err && python bench.py --no-cpu-baseline --precision bf16 > gpurun_out/b_bf.json 2>>gpurun_out/b_x3.err
python - <<PY
import json
for f in ["gpurun_out/b_x3.json","gpurun_out/b_bf.json"]:
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(j["dtype"], round(j["ms_per_step"],4), {k:round(v*1000,1) for k,v in j["kernels_ms"].items()}, j["loss_mean"])
PY
export VPC_LIB=$PWD/vae-posterior-consistency_amd/csrc/libvpc_hip_ablate.so VPC_DEBUG=64 VPC_DEBUG_ENC=64
python tools/stamps_small.py 65536 bf16x3 > gpurun_out/stamps_bf16x3_b.log 2>&1 && python tools/stamps_small.py 65536 bf16 > gpurun_out/stamps_bf16_b.log 2>&1
grep -B9 "step 1" gpurun_out/stamps_bf16x3_b.log | grep "wave 0" | grep -v "blk 0 \|enc_fwd"; grep -B9 "step 1" gpurun_out/stamps_bf16_b.log | grep "wave 0" | grep -v "blk 0 \|enc_fwd"
