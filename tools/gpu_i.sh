#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=r02i
timeout 300 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py tests/test_dp_gpu.py -k "split_generator or dp_" > $O/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/${TAG}_tests.log | cut -c1-220; grep -n "Error" $O/${TAG}_tests.log | cut -c1-500 | head -6
timeout 300 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/${TAG}_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['traffic'], d['parity_mode']['ms_per_step'])"
(cd /tmp && SSR_OVERLAP_D=0 SSR_G_SPLIT=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/prof_$TAG.log 2>&1)
F=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $O/${TAG}_kernel_stats_serial.csv && head -6 $F | cut -c1-150
tail -1 /tmp/prof_$TAG.log | cut -c1-200
echo "== serial step"; SSR_OVERLAP_D=0 SSR_G_SPLIT=0 python bench.py --no-cpu-baseline --no-parity-mode --no-roofline --blocks-timed 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'])"
