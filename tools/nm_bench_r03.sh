#!/bin/bash
# MNAR step (config 3), plain bf16: layer-fused decoder kernel vs the GEMM chain (VPC_NMDEC=0), eager and graph replay
mkdir -p gpurun_out/r03
out=gpurun_out/r03/${1:-nmdec_bench}.jsonl; rm -f $out
for B in ${2:-128 8192 65536}; do
  for env in "" "VPC_NMDEC=0"; do
    for g in "" "--graph"; do
      [ "$B" != "128" ] && [ -n "$g" ] && continue
      env $env timeout -k 10 200 python tools/bench_mnar.py --batch $B --precision bf16 --no-cpu --steps 200 --warmup 30 $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d['form']='gemm' if '$env' else 'fused'; print(json.dumps(d))" >> $out || exit 1
    done
  done
done
python - <<PY
import json
for l in open("$out"):
    d=json.loads(l); print(d['form'], d['config']['workload'].split()[2], 'graph' if d['config']['graph'] else 'eager', '%.4f ms'%d['ms_per_step'])
PY
