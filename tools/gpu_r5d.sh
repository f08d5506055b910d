#!/bin/bash
# r05d: store phase inside the MFMA stream: probe, parity, step
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for a in "32 64 32" "32 160 32" "32 192 64"; do tools/x3_probe $a; done > $O/r05d_x3_probe.txt 2>&1; cat $O/r05d_x3_probe.txt | grep -v skew
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_parity.py -m gpu -x -k "conv_layer_fwd_dgrad_wgrad and fp32x3" > $O/r05d_tests.log 2>&1; echo "pytest conv rc=$?"; tail -2 $O/r05d_tests.log
for V in "SSR_X3_PIPE=0" "SSR_X3_PIPE=1" "SSR_X3_PIPE2=1"; do
  echo "== $V"; env $V timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 10 --warmup 3 --blocks-timed 1 2> $O/r05d_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('max_rel_err_vs_oracle'), json.dumps(d['kernel_time_breakdown_ms']))"
done
