#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_rdb_tile.py tests/test_gpu_rdb_stress.py tests/test_gpu_parity.py 2>&1 | tail -2
bash tools/ab.sh ab/after_pro.so ab/after_old2.so --no-parity-mode --frames 1 --batch 16 2>&1 | tail -8
