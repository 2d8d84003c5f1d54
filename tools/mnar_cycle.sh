#!/bin/bash
# GPU cycle for the MNAR path: parity tests, then the step bench at three batch sizes with the per-launch breakdown
timeout -k 10 500 python -m pytest tests/test_notmiwae_gpu.py -x -q -m gpu > gpurun_out/nm_test.log 2>&1; tail -4 gpurun_out/nm_test.log
grep -q passed gpurun_out/nm_test.log && ! grep -q failed gpurun_out/nm_test.log || exit 1
timeout -k 10 300 python tools/bench_mnar.py --batch 128 --no-cpu > gpurun_out/mnar_b128_nt.json 2>> gpurun_out/mnar_err.log &&
timeout -k 10 300 python tools/bench_mnar.py --batch 128 --no-cpu --graph --steps 200 > gpurun_out/mnar_b128_graph.json 2>> gpurun_out/mnar_err.log &&
timeout -k 10 300 python tools/bench_mnar.py --batch 8192 --steps 20 --warmup 3 --timers --no-cpu > gpurun_out/mnar_b8192.json 2>> gpurun_out/mnar_err.log &&
timeout -k 10 300 python tools/bench_mnar.py --batch 65536 --steps 5 --warmup 2 --timers --no-cpu > gpurun_out/mnar_b65536.json 2>> gpurun_out/mnar_err.log
python - <<EOF2
import json
for f in ["mnar_b128_nt","mnar_b128_graph","mnar_b8192","mnar_b65536"]:
    j=json.load(open("gpurun_out/%s.json"%f)); print(f, round(j["ms_per_step"],3), round(j["roofline"]["frac"],3), {k:round(v,3) for k,v in j.get("kernels_ms",{}).items()})
EOF2
