#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=r02c
B="python bench.py --no-cpu-baseline --no-parity-mode --no-roofline --blocks-timed 3"
for i in 1 2; do
  echo "== base";      $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'])"
  echo "== overlap_d"; SSR_OVERLAP_D=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'])"
  echo "== wgrad T3=64"; SSR_WGRAD_T3=64 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'])"
done
echo "== overlap + T3=64"; SSR_OVERLAP_D=1 SSR_WGRAD_T3=64 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'])"
echo "== perceptual"; $B --perceptual 2>$O/${TAG}_percep.err | tee $O/${TAG}_bench_perceptual.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks'])" || tail -5 $O/${TAG}_percep.err
echo "== infer"; python tools/infer_bench.py 64 2>/dev/null | tee $O/${TAG}_infer.txt | tail -1
SSR_OVERLAP_D=1 timeout 400 python -m pytest -q -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_perceptual.py -k "train_step or gated or old_hr or perceptual or resume or validation or metrics or quantize or learning" -x > $O/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; tail -5 $O/${TAG}_tests.log | cut -c1-300; grep -n "Error" $O/${TAG}_tests.log | cut -c1-500 | head
