#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 500 python -m pytest -q -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -k "conv_layer or golden or stride2 or (every_layer and discriminator and 3-32) or (every_layer and generator and 3-4)" > $O/r02p_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/r02p_tests.log | cut -c1-220; grep -n "Error" $O/r02p_tests.log | cut -c1-500 | head -6
