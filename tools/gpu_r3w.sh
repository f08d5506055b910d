#!/bin/bash
L=satlas_super_resolution_amd/libssr_hip.so
for v in ab/rq1.so ab/rq2.so; do cp $v $L; echo "== $v"; timeout 200 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_rdb_tile.py 2>&1 | tail -1; done
true
