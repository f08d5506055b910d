#!/bin/bash
# lesson 36: which wait in front of the producers' refill loads hides the race (library variants under ab/, not committed)
L=satlas_super_resolution_amd/libssr_hip.so
cp $L /tmp/keep.so
for v in "$@"; do cp $v $L; echo "== $v"; timeout 200 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_rdb_tile.py 2>&1 | tail -1; done
cp /tmp/keep.so $L
