#!/bin/bash
# round 6: the one-pass 4x4 stride-2 weight-gradient kernel (wgrad_x3_k4_kernel) - parity, then A/B of the step in one call
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_wgrad_x3.py -m gpu -x 2>&1 | tail -8
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -m gpu -k "discriminator and fp32h" 2>&1 | tail -3
bash tools/ab_envn.sh "SSR_X3_WGRAD_FUSED4=0" "SSR_X3_WGRAD_FUSED4=1" 2>&1 | tail -6
