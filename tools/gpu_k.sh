#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
SSR_D_AHEAD=1 timeout 400 python -m pytest -q -p no:cacheprovider -x tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py tests/test_gpu_perceptual.py -k "train_step_golden or train_step_vs_oracle or (full_depth and train_step and 24) or shipped_loss" > $O/r02k_tests.log 2>&1
echo "pytest rc=$?"; tail -5 $O/r02k_tests.log | cut -c1-250; grep -n "Error" $O/r02k_tests.log | cut -c1-500 | head -8
B="python bench.py --no-cpu-baseline --no-parity-mode --no-roofline --blocks-timed 3"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_blocks"])'
for i in 1 2; do
  echo "== base"; $B 2>/dev/null | python -c "$P"
  echo "== d_ahead"; SSR_D_AHEAD=1 $B 2>$O/r02k_ahead.err | python -c "$P" || tail -5 $O/r02k_ahead.err
done
echo "== d_ahead, cfg1"; SSR_D_AHEAD=1 $B --frames 1 --batch 16 2>/dev/null | python -c "$P"; echo "== base, cfg1"; $B --frames 1 --batch 16 2>/dev/null | python -c "$P"
