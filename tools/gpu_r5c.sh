#!/bin/bash
# r05c: the split-mode big-tile kernel: parity, then the step with / without it
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py -m gpu -x > $O/r05c_tests.log 2>&1; echo "pytest x3 rc=$?"; tail -15 $O/r05c_tests.log | cut -c1-300
for V in 0 1; do
  echo "== SSR_X3_BIGTILE=$V"; SSR_X3_BIGTILE=$V SSR_BENCH_LAYER_DUMP=$O/r05c_layers_big$V.txt timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 10 --warmup 3 --blocks-timed 1 2> $O/r05c_bench$V.err | tee $O/r05c_bench_big$V.json | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('max_rel_err_vs_oracle'), json.dumps(d['kernel_time_breakdown_ms']), json.dumps(d['roofline']))"
done
tail -3 $O/r05c_bench1.err
