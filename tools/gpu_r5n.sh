#!/bin/bash
# r05n: discriminator conv3 (16x16 grid) through the space-to-depth big-tile kernel in fp32x3: the D tests at the benchmarked shapes, the step
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_baseline_shapes.py tests/test_gpu_deterministic.py -k "fp32x3 or x3" > $O/r05n_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r05n_tests.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5 --blocks-timed 2 2> $O/r05n_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['ms_per_step_blocks'], d.get('max_rel_err_vs_oracle'), json.dumps(d['kernel_time_breakdown_ms']))"
