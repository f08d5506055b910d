#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_rdb_tile.py tests/test_gpu_rdb_stress.py tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -k "not full_size" 2>&1 | tail -2
bash tools/ab.sh ab/before_pro.so ab/after_pro.so --no-parity-mode 2>&1 | tail -8
