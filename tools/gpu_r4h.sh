#!/bin/bash
# round 4, visit h: new tests (deterministic mode, 2-rank train(), USM fixture, per-call switches) + cost of the deterministic mode
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
PYT="python -m pytest -q -p no:cacheprovider -m gpu"
timeout 1500 $PYT tests/test_gpu_deterministic.py tests/test_gpu_rdb_tile.py tests/test_gpu_rdb_stress.py tests/test_dp_gpu.py tests/test_gpu_boundary.py tests/test_gpu_parity.py --durations=8 > $O/r04h_tests.log 2>&1
tail -25 $O/r04h_tests.log
for det in 0 1; do
  echo "== SSR_DETERMINISTIC=$det"
  SSR_DETERMINISTIC=$det python bench.py --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/r04h_det_cost.txt
