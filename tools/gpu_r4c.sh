#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
{
for v in sel1 il5 il5rq4 il5rq2; do echo "== $v"; timeout 60 tools/rdbt_x_$v check | tail -1; timeout 60 tools/rdbt_x_$v time32 | grep "tile 16"; done
echo "== p_il5"; timeout 60 tools/rdbt_x_p_il5 probe 32
} > $O/r04c_probes.log 2>&1
cat $O/r04c_probes.log
