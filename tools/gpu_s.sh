#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 500 python -m pytest -q -p no:cacheprovider tests/test_gpu_boundary.py tests/test_gpu_parity.py -k "not conv_layer and not weight_stationary and not big_tile and not thin_output and not stride2" > $O/r02s_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/r02s_tests.log | cut -c1-220; grep -n "Error" $O/r02s_tests.log | cut -c1-600 | head -8
