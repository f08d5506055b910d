#!/bin/bash
timeout 1200 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -x -m gpu -k "vs_reference_class and fp32f" -s 2>&1 | grep -E "fp32f|passed|failed|Error|assert" | tail -40
python bench.py --dtype fp32f --steps 10 --warmup 3 --no-cpu-baseline --no-legs --blocks-timed 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fp32f', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['frac'],3), d.get('max_rel_err_vs_oracle')); print(d['kernel_time_breakdown_ms'])"
