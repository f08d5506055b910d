#!/bin/bash
# serial-mode kernel stats of the final build: one launch chain, every kernel has the chip to itself (what bench.py's roofline measures)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && SSR_OVERLAP_D=0 SSR_G_SPLIT=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ser -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > $O/r03x_serial_bench.json 2> /tmp/prof_ser.log)
F=$(find /tmp/prof_ser -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $O/r03x_kernel_stats_serial.csv && head -8 $F | cut -c1-170
python -c "
import json; d=json.loads(open('$O/r03x_serial_bench.json').read().strip().splitlines()[-1]); print('serial step', d['ms_per_step'], d['value'])"
python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('default step', d['ms_per_step'], d['value'], r['kernel'], r['frac'], r['avg_launch_us'], 'traffic', r['traffic'])"
