#!/bin/bash
timeout 600 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_parity.py -k "bilinear" 2>&1 | tail -3
for ord in heavy xcd; do
  SSR_WGRAD_ORDER=$ord python bench.py --no-parity-mode --no-cpu-baseline > gpurun_out/r03u_bench_$ord.json 2> gpurun_out/r03u_bench_$ord.err; echo "order=$ord rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/r03u_bench_$ord.json')); print(d['ms_per_step'], d['value']); print({k:round(v,3) for k,v in d['kernel_time_breakdown_ms'].items() if 'wgrad' in k})"
done
python bench.py --dtype fp32x3 --no-parity-mode --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fp32x3', d['ms_per_step'], d['value']); print({k:round(v,3) for k,v in sorted(d['kernel_time_breakdown_ms'].items(), key=lambda kv:-kv[1])[:12]})"
