#!/bin/bash
# round 6, visit u: prologue variants of the register-tiled body kernel (weight steps before the first-chunk wait, producer priority), 8- vs 4-row tiles at N = 32
O=gpurun_out; mkdir -p $O
{
for v in base wpre3 wpre1 prio1 w3p1 w3p2; do
  for a in "32 64 32" "32 128 32" "32 192 64"; do echo "== variant $v TH=8"; timeout 60 tools/x3r_x_$v $a; done
done
for v in base wpre3; do
  for a in "32 64 32" "32 128 32" "32 160 32" "32 192 64"; do echo "== variant $v TH=4"; SSR_X3_REGTILE_TH=4 SSR_X3_REGTILE_NT2=0 timeout 60 tools/x3r_x_$v $a; done
done
} > $O/r06u_x3r_prologue.txt 2>&1
grep -E "variant|avg launch|whole|first chunk|first loads" $O/r06u_x3r_prologue.txt
