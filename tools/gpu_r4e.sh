#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
{
for v in c5x0 c5x; do echo "== $v"; timeout 120 tools/rdbt_x_$v check | tail -3; timeout 60 tools/rdbt_x_$v time32 | grep "tile 16"; done
echo "== p_c5x"; timeout 60 tools/rdbt_x_p_c5x probe 32
} > $O/r04e_probes.log 2>&1
cat $O/r04e_probes.log
