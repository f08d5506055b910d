#!/bin/bash
# round 6, visit v: per-kernel breakdown of the fp32f mode (exact forward, split backward) beside fp32x3
O=gpurun_out; mkdir -p $O
python bench.py --dtype fp32f --no-legs --no-cpu-baseline --blocks-timed 1 > $O/r06v_bench_fp32f.json 2> $O/r06v_bench_fp32f.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r06v_bench_fp32f.json').read().strip().splitlines()[-1])
print(d['dtype'], d['ms_per_step'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'))
print(json.dumps(d.get('kernel_time_breakdown_ms'), indent=1))
P
