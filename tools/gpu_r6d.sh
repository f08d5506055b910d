#!/bin/bash
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py tests/test_gpu_x3_stress.py -x -m gpu 2>&1 | tail -4
bash tools/ab_envn.sh "SSR_X3_REGTILE=0" "SSR_X3_REGTILE=1" "SSR_G_SPLIT=2" 2>&1 | tail -8
