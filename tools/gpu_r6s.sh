#!/bin/bash
timeout 1500 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py -x -m gpu -k "exact" 2>&1 | tail -4
