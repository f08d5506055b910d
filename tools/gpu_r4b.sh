#!/bin/bash
# round 4, visit b: in-block stage times of the dense block (SSR_PROBE builds), with / without hand-over, at 256 and 64 workgroups
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
{
for v in p_sel1 p_sel0 p_nosync p_noprod; do for n in 32 8; do echo "== $v N=$n"; timeout 60 tools/rdbt_x_$v probe $n; done; done
} > $O/r04b_probes.log 2>&1
cat $O/r04b_probes.log
