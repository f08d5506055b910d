#!/bin/bash
# r05l: fp32x3 big-tile kernel, epilogue transposed through LDS and branch-free (template flags): parity tests, the probe again, the step
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_conv_x3.py tests/test_gpu_parity.py -k "not ring" > $O/r05l_tests.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r05l_tests.log | cut -c1-400
{
for A in "32 64 64 128 128 0" "32 64 64 128 128 1" "32 64 64 128 128 2" "32 128 64 128 128 0" "32 256 128 64 64 0" "32 512 256 32 32 0"; do
  timeout 120 tools/bigx3_probe $A
done
} > $O/r05l_bigx3_probe.txt 2>&1
grep -E "^N=|epilogues|k-steps" $O/r05l_bigx3_probe.txt
timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5 --blocks-timed 2 2> $O/r05l_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['ms_per_step_blocks'], d.get('max_rel_err_vs_oracle'), json.dumps(d['kernel_time_breakdown_ms']))"
