python -m pytest tests/test_bf16.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/t_bf.log 2>&1; tail -2 gpurun_out/t_bf.log
for p in bf16x3 bf16 f32; do python bench.py --no-cpu-baseline --precision $p > gpurun_out/b_$p.json 2>>gpurun_out/b_x3.err; done
python - <<PY
import json
for p in ["bf16x3","bf16","f32"]:
    j=json.loads(open(f"gpurun_out/b_{p}.json").read().strip().splitlines()[-1]); print(j["dtype"], round(j["ms_per_step"],4), {k:round(v*1000,1) for k,v in j["kernels_ms"].items()}, j["loss_mean"])
PY
