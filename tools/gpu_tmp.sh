#!/bin/bash
cd /root/repo
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_parity.py -m gpu -x -k "weight_stationary or discriminator or conv_layer or generator or golden" 2>&1 | tail -4
bash tools/ab.sh ab/base.so ab/new.so --no-parity-mode --no-roofline 2>&1 | tail -8
bash tools/ab_env.sh SSR_CONV_WS21 1 3 --no-parity-mode --no-roofline 2>&1 | tail -8
