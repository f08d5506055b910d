#!/bin/bash
# round 6: the one-pass 4x4 stride-2 weight-gradient kernel (wgrad_x3_k4_kernel; paired dY blocks) - parity, then A/B of the step in one call
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_wgrad_x3.py -m gpu 2>&1 | tail -8
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -m gpu -k "discriminator and fp32h" 2>&1 | tail -3
for r in 1 2; do for v in "SSR_X3_WGRAD_FUSED4=0" "SSR_WGRAD_PAIR=0" "SSR_WGRAD_PAIR=1"; do
  echo -n "$v  "; env $v python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
done; done 2>&1 | tee $O/r06k4_wgrad4_ab.txt
python bench.py --no-cpu-baseline --no-legs --blocks-timed 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k:v for k,v in d['roofline_by_kernel'].items() if 'wgrad' in k})" | tee -a $O/r06k4_wgrad4_ab.txt
