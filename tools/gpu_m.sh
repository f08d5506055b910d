#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --dtype fp32x3 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline --blocks-timed 0"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'
for thr in 384 0 100000; do echo "== SSR_X3_SMALL=$thr"; SSR_X3_SMALL=$thr $B 2>/dev/null | python -c "$P"; done
timeout 400 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -k "infer_grid or (every_layer and generator and (16 or fp32))" > $O/r02m_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/r02m_tests.log | cut -c1-220; grep -n "Error" $O/r02m_tests.log | cut -c1-500 | head -6
