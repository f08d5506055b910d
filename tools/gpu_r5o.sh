#!/bin/bash
# r05o: thin-output VALU kernel in the fp32 modes: its tests, the conv-layer parity tests in fp32 / fp32x3, the step
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_conv_x3.py tests/test_gpu_parity.py -k "thin or conv_layer" > $O/r05o_tests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r05o_tests.log | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5 --blocks-timed 2 2> $O/r05o_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['ms_per_step_blocks'], d.get('max_rel_err_vs_oracle'), json.dumps(d['kernel_time_breakdown_ms']))"
