#!/bin/bash
timeout 600 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_parity.py 2>&1 | tail -2
for bal in 0 1; do
  SSR_WGRAD_BALANCE=$bal python bench.py --no-parity-mode --no-cpu-baseline > gpurun_out/r03q_bench_bal$bal.json 2> gpurun_out/r03q_bench_bal$bal.err; echo "bal=$bal rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/r03q_bench_bal$bal.json')); print(d['ms_per_step'], d['value']); print({k:round(v,3) for k,v in d['kernel_time_breakdown_ms'].items() if 'wgrad' in k})"
done
