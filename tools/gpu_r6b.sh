#!/bin/bash
O=gpurun_out; mkdir -p $O
{
for v in "$@"; do
  for a in "32 64 32" "32 128 32" "32 192 64"; do echo "== variant $v"; timeout 60 tools/x3r_x_$v $a; done
done
} > $O/r06b_x3r_variants.txt 2>&1
grep -E "variant|avg launch|item 2|item 3|whole|first chunk|sum \+" $O/r06b_x3r_variants.txt
