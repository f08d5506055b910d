#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_rdb_tile.py tests/test_gpu_rdb_stress.py 2>&1 | tail -2
bash tools/ab.sh ab/c5x0.so ab/c5x1.so --no-parity-mode 2>&1 | tail -8
