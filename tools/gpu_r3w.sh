#!/bin/bash
L=satlas_super_resolution_amd/libssr_hip.so
cp $L /tmp/keep.so
for v in ab/fence_arith.so ab/fence_sel.so; do cp $v $L; echo "== $v"; timeout 200 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_rdb_tile.py 2>&1 | tail -1; done
cp /tmp/keep.so $L
timeout 60 tools/rdbt_check time32 | grep "tile 16"
