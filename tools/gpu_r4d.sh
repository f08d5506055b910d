#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
{
for v in sel1 nowload; do echo "== $v"; timeout 60 tools/rdbt_x_$v time32 | grep "tile 16"; done
echo "== p_nowload"; timeout 60 tools/rdbt_x_p_nowload probe 32
} > $O/r04d_probes.log 2>&1
cat $O/r04d_probes.log
