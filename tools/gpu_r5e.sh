#!/bin/bash
# r05e: the ring kernel of the fp32x3 body: probe, parity, step
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for a in "32 64 32" "32 160 32" "32 192 64"; do timeout 60 tools/x3q_probe $a; done > $O/r05e_x3q_probe.txt 2>&1; cat $O/r05e_x3q_probe.txt
timeout 600 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py -m gpu -x -k "ring" > $O/r05e_tests.log 2>&1; echo "pytest ring rc=$?"; tail -4 $O/r05e_tests.log | cut -c1-400
for V in 0 1; do
  echo "== SSR_X3_RING=$V"; SSR_X3_RING=$V timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 10 --warmup 3 --blocks-timed 1 2> $O/r05e_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('max_rel_err_vs_oracle'), json.dumps(d['kernel_time_breakdown_ms']))"
done
