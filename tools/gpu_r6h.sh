#!/bin/bash
timeout 300 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py -x -m gpu -k "chain" 2>&1 | tail -12
timeout 120 python tools/chain_time.py 32 200 2>&1 | tail -5
timeout 120 python tools/chain_time.py 16 200 2>&1 | tail -5
