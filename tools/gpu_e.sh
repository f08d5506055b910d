#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=r02e
timeout 300 python -m pytest -q -p no:cacheprovider tests/test_gpu_perceptual.py "tests/test_gpu_baseline_shapes.py::test_train_step_full_depth_vs_oracle" -k "perceptual or maxpool or (fp32x3 and 96)" > $O/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; tail -6 $O/${TAG}_tests.log | cut -c1-260; grep -n "Error" $O/${TAG}_tests.log | cut -c1-700 | head -12
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o k -- python $R/bench.py --dtype fp32x3 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/prof_$TAG.log 2>&1)
F=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $O/${TAG}_kernel_stats_fp32x3.csv && head -16 $F | cut -c1-170
tail -2 /tmp/prof_$TAG.log | cut -c1-300
