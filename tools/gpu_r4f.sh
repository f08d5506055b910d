#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
{
for r in 1 2; do for v in c5x c5x_prio10 c5x_prio11 c5x_prio21 c5x_sleep1 c5x_rq4 c5x_pf3; do echo "== $v"; timeout 60 tools/rdbt_x_$v time32 | grep "tile 16"; done; done
} > $O/r04f_probes.log 2>&1
cat $O/r04f_probes.log
