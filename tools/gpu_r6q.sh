#!/bin/bash
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py -x -m gpu -k "exact or regtile" 2>&1 | tail -4
for r in 1 2; do for v in 0 1; do echo -n "fp32 SSR_F32_REGTILE=$v  "; SSR_F32_REGTILE=$v python bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-legs --blocks-timed 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; done; done
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -x -m gpu -k "fp32 and not fp32x3" 2>&1 | tail -4
