#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py tests/test_gpu_x3_stress.py -x -m gpu 2>&1 | tail -4
for a in "32 192 64" "32 192 64 32 32 1" "32 64 64 32 32 1" "16 192 64"; do SSR_X3_REGTILE_NT2=wave tools/x3r_x_w8 $a | grep -E "avg launch|sum \+|whole"; tools/x3r_x_w8 $a | grep -E "avg launch|sum \+|whole|item 3"; done
bash tools/ab_envn.sh "SSR_X3_REGTILE_NT2=wave" "SSR_X3_REGTILE_NT2=8" 2>&1 | tail -4
