#!/bin/bash
true
timeout 600 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_parity.py 2>&1 | tail -2
python bench.py --no-parity-mode --no-cpu-baseline > gpurun_out/r03s_bench.json 2> gpurun_out/r03s_bench.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r03s_bench.json')); print(d['ms_per_step'], d['value']); print({k:round(v,3) for k,v in d['kernel_time_breakdown_ms'].items() if 'wgrad' in k})"
