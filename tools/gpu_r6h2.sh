#!/bin/bash
timeout 600 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py -m gpu -k "overflow or accuracy_over" 2>&1 | tail -5
