#!/bin/bash
# A/B of two builds of libssr_hip.so inside ONE gpurun call (box-to-box variance is ~5 %): tools/ab.sh ab/base.so ab/new.so [bench args]
A=$1; B=$2; shift 2
L=satlas_super_resolution_amd/libssr_hip.so
cp $L /tmp/keep.so
for r in 1 2; do
  for v in $A $B; do
    cp $v $L
    echo "== $v"; python bench.py --steps 30 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
cp /tmp/keep.so $L
