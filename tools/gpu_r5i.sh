#!/bin/bash
# r05i: where the fp32x3 big-tile kernel's time goes (tools/bigx3_probe): the discriminator's 128^2 / 64^2 / 32^2 layers
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
{
for A in "32 64 64 128 128 0" "32 64 64 128 128 1" "32 64 64 128 128 2" "32 128 64 128 128 0" "32 256 128 64 64 0" "32 512 256 32 32 0" "16 64 64 128 128 0"; do
  timeout 120 tools/bigx3_probe $A
done
} > $O/r05i_bigx3_probe.txt 2>&1
cat $O/r05i_bigx3_probe.txt
