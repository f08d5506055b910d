#!/bin/bash
timeout 120 tools/rdbt_check check | tail -4
timeout 120 tools/rdbt_check time32 | tail -6
python bench.py --no-parity-mode --no-cpu-baseline > gpurun_out/r03r_bench.json 2> gpurun_out/r03r_bench.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r03r_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us']); print({k:round(v,3) for k,v in d['kernel_time_breakdown_ms'].items()})"
