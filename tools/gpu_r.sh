#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest -q -p no:cacheprovider tests/test_gpu_boundary.py -k "test_pipeline or validation or training_loop" > $O/r02r_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/r02r_tests.log | cut -c1-220; grep -n "Error" $O/r02r_tests.log | cut -c1-600 | head -8
