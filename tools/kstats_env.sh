#!/bin/bash
# per-kernel time under an environment setting: tools/kstats_env.sh "VAR=x" tag -> gpurun_out/kstats_<tag>.csv
ENVV=$1; TAG=$2
R=$(pwd); export TMPDIR=/tmp; mkdir -p $R/gpurun_out
(cd /tmp && env $ENVV rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_$TAG.log 2>&1)
F=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp $F $R/gpurun_out/kstats_$TAG.csv
