#!/bin/bash
timeout 600 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_parity.py 2>&1 | tail -3
for fl in 1 0; do
  SSR_BILINEAR_FLAT=$fl python bench.py --no-parity-mode --no-cpu-baseline > gpurun_out/r03t_bench_flat$fl.json 2> gpurun_out/r03t_bench_flat$fl.err; echo "flat=$fl rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/r03t_bench_flat$fl.json')); print(d['ms_per_step'], d['value']); print({k:round(v,3) for k,v in d['kernel_time_breakdown_ms'].items() if 'bilinear' in k})"
done
