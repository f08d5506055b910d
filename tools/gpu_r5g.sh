#!/bin/bash
# r05g: batch-independent kernel choice, one-launch split, ssr_fill instead of ATen fills: the affected tests, then the step
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_conv_x3.py tests/test_gpu_deterministic.py tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py -k "x3 or deterministic or infer_grid_tile or train_step_golden or conv_layer" > $O/r05g_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r05g_tests.log | cut -c1-300
for V in heavy xcd; do
  echo "== SSR_WGRAD_ORDER=$V"; SSR_WGRAD_ORDER=$V timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5 --blocks-timed 2 2> $O/r05g_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['ms_per_step_blocks'], d.get('max_rel_err_vs_oracle'), json.dumps(d['kernel_time_breakdown_ms']))"
done
for V in heavy xcd; do
  echo "== bf16 SSR_WGRAD_ORDER=$V"; SSR_WGRAD_ORDER=$V timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-legs --steps 30 --warmup 10 --blocks-timed 2 --no-roofline 2> $O/r05g_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['ms_per_step_blocks'])"
done
