#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
{
for v in c5x pro; do echo "== $v"; timeout 120 tools/rdbt_x_$v check | tail -2; timeout 60 tools/rdbt_x_$v time32 | grep "tile 16"; done
echo "== p_pro"; timeout 60 tools/rdbt_x_p_pro probe 32
} > $O/r04n_probes.log 2>&1
cat $O/r04n_probes.log
