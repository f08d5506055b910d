#!/bin/bash
# r05b: the deep-pipeline split kernel: conv parity in fp32x3, layer-local checks, A/B of the step against SSR_X3_PIPE=0
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_parity.py -m gpu -x -k "conv_layer_fwd_dgrad_wgrad and fp32x3" > $O/r05b_tests.log 2>&1; echo "pytest conv rc=$?"; tail -3 $O/r05b_tests.log
for V in 0 1; do
  echo "== SSR_X3_PIPE=$V"; SSR_X3_PIPE=$V SSR_BENCH_LAYER_DUMP=$O/r05b_layers_pipe$V.txt timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 10 --warmup 3 --blocks-timed 1 2> $O/r05b_bench$V.err | tee $O/r05b_bench_pipe$V.json | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['max_rel_err_vs_oracle'] if 'max_rel_err_vs_oracle' in d else '', json.dumps(d['kernel_time_breakdown_ms']))"
done
timeout 1200 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -m gpu -x -s -k "fp32x3 and (every_layer or forward)" >> $O/r05b_tests.log 2>&1; echo "pytest layers rc=$?"; grep -E "passed|failed|layerwise" $O/r05b_tests.log | tail -8
