#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 100 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -k "(vs_reference_class and full_d3) or (train_step_full_depth and fp32x3-3-False)" > $O/r02v_tests.log 2>&1
echo "pytest rc=$?"; tail -3 $O/r02v_tests.log | cut -c1-220; grep -n "Error" $O/r02v_tests.log | cut -c1-500 | head -5
