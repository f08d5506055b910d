#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -s -k "vs_reference_class" > $O/r02u_tests.log 2>&1
echo "pytest rc=$?"; grep -E "^\[fp32|passed|failed" $O/r02u_tests.log | cut -c1-220; grep -n "Error" $O/r02u_tests.log | cut -c1-700 | head -10
