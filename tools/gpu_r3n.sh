#!/bin/bash
for b in wgrad_body_probe wgrad_body_probe_pf8 wgrad_body_probe_NOLOAD; do echo "== $b"; timeout 60 tools/$b 69 2 0 1 | head -2; done
timeout 600 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_parity.py 2>&1 | tail -2
python bench.py --no-parity-mode --no-cpu-baseline > gpurun_out/r03p_bench.json 2> gpurun_out/r03p_bench.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r03p_bench.json')); print(d['ms_per_step'], d['value']); print({k:round(v,3) for k,v in d['kernel_time_breakdown_ms'].items() if 'wgrad' in k})"
