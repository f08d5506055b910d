#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 200 python -m pytest -q -s -p no:cacheprovider tests/test_gpu_baseline_shapes.py -k "split_generator" 2>&1 | grep -E "^\[split|passed|failed"
