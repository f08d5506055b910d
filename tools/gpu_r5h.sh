#!/bin/bash
# r05h: one-pass fp32x3 weight gradient (csrc/wgrad_x3.hip): its tests, the conv-layer parity tests, then the step with and without it
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_wgrad_x3.py tests/test_gpu_parity.py -k "wgrad or conv_layer" > $O/r05h_tests.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r05h_tests.log | cut -c1-400
for V in 1 0; do
  echo "== SSR_X3_WGRAD_FUSED=$V"; SSR_X3_WGRAD_FUSED=$V timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5 --blocks-timed 2 2> $O/r05h_bench$V.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['ms_per_step_blocks'], d.get('max_rel_err_vs_oracle'), json.dumps(d['kernel_time_breakdown_ms']))"
done
