#!/bin/bash
# round 6, visit y: the fp16-split forward arithmetic (mode fp32h) - kernel parity, the full-size gate tests, bench beside fp32x3 / fp32f
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py -m gpu -k "fp16" -x 2>&1 | tail -15
timeout 1500 python -m pytest -q -p no:cacheprovider -s tests/test_gpu_baseline_shapes.py -m gpu -k "fp32h" 2>&1 | grep -E "fp32h|passed|failed|Error|assert" | tail -80 > $O/r06y_fp32h_fullsize.txt; tail -30 $O/r06y_fp32h_fullsize.txt
python bench.py --dtype fp32h --no-legs --no-cpu-baseline --blocks-timed 1 > $O/r06y_bench_fp32h.json 2> $O/r06y_bench_fp32h.err || tail -5 $O/r06y_bench_fp32h.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r06y_bench_fp32h.json').read().strip().splitlines()[-1])
print(d['dtype'], d['ms_per_step'], d['value'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), d.get('max_rel_err_vs_oracle'))
print(json.dumps(d.get('kernel_time_breakdown_ms'), indent=1))
P
