#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest -q -p no:cacheprovider -m gpu -s tests/test_dp_gpu.py tests/test_gpu_infer_scripts.py tests/test_gpu_rdb_tile.py "tests/test_gpu_boundary.py::test_model_plugin_against_the_unmodified_reference_method" > $O/r03i_tests.log 2>&1; echo "pytest rc=$?"; grep -E "^\[|passed|failed|Error|error|assert" $O/r03i_tests.log | cut -c1-240 | head -40
