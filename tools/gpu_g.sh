#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-parity-mode --no-roofline --blocks-timed 3"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_blocks"])'
for i in 1 2; do
  echo "== single chain"; $B 2>/dev/null | python -c "$P"
  echo "== split2"; SSR_G_SPLIT=2 $B 2>$O/r02g_split.err | python -c "$P" || tail -5 $O/r02g_split.err
  echo "== split4"; SSR_G_SPLIT=4 $B 2>$O/r02g_split.err | python -c "$P" || tail -5 $O/r02g_split.err
done
echo "== split2 + wgrad T3=32"; SSR_G_SPLIT=2 SSR_WGRAD_T3=32 $B 2>/dev/null | python -c "$P"
timeout 300 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -k "split_generator" > $O/r02g_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/r02g_tests.log | cut -c1-300; grep -n "Error" $O/r02g_tests.log | cut -c1-400 | head -5
