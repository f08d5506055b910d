#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
PYT="python -m pytest -q -p no:cacheprovider -m gpu"
timeout 1500 $PYT tests/test_gpu_deterministic.py tests/test_dp_gpu.py "tests/test_gpu_boundary.py::test_model_plugin_against_the_unmodified_reference_method" tests/test_gpu_boundary.py::test_training_loop_on_the_miniature_dataset > $O/r04i_tests.log 2>&1
grep -E "^E  |passed|failed|^FAILED" $O/r04i_tests.log | head -30
