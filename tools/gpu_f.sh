#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-parity-mode --no-roofline --blocks-timed 3"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_blocks"])'
for i in 1 2; do
  for c in 1 3 6 12; do echo "== chunks $c"; SSR_WGRAD_CHUNKS=$c $B 2>/dev/null | python -c "$P"; done
done
SSR_WGRAD_CHUNKS=4 timeout 300 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py -k "(every_layer and generator and 24-32) or train_step_golden" > $O/r02f_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/r02f_tests.log | cut -c1-200
