#!/bin/bash
for a in "69 0 0 0" "69 2 0 1" "69 1 0 1" "69 0 0 1" "23 2 0 1"; do timeout 60 tools/wgrad_body_probe $a; done
timeout 600 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_parity.py tests/test_layerwise_oracle.py 2>&1 | tail -5
for pr in 0 1; do
  SSR_WGRAD_PAIR=$pr python bench.py --no-parity-mode --no-cpu-baseline > gpurun_out/r03m_bench_pair$pr.json 2> gpurun_out/r03m_bench_pair$pr.err; echo "pair=$pr rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/r03m_bench_pair$pr.json')); print('pair$pr', d['ms_per_step'], d['value']); print({k:round(v,3) for k,v in d['kernel_time_breakdown_ms'].items() if 'wgrad' in k})"
done
