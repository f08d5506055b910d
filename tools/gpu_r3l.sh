#!/bin/bash
# wgrad item order A/B: same box, bf16 step only
O=gpurun_out; mkdir -p $O
for ord in heavy xcd; do
  SSR_WGRAD_ORDER=$ord python bench.py --no-parity-mode --no-cpu-baseline > $O/r03l_bench_$ord.json 2> $O/r03l_bench_$ord.err; echo "$ord rc=$?"
  python -c "
import json; d=json.load(open('$O/r03l_bench_$ord.json')); print('$ord', d['ms_per_step'], d['value']); print({k:round(v,3) for k,v in d['kernel_time_breakdown_ms'].items() if 'wgrad' in k})"
done
timeout 300 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_parity.py -k "wgrad or step or grad" 2>&1 | tail -3
