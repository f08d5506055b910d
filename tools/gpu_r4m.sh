#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for thr in 1e-4 1e-3; do
echo "== SSR_X3_FIX_THR=$thr"
SSR_X3_FIX_THR=$thr timeout 1500 python -m pytest -q -p no:cacheprovider -m gpu -s tests/test_gpu_baseline_shapes.py -k "full_size and fp32x3 and not nofix" > $O/r04m_fullsize_$thr.log 2>&1
grep -E "^\[fp32|passed|failed|^E  " $O/r04m_fullsize_$thr.log | cut -c1-200 | grep -E "dx|conv_first|conv0|masked|largest|passed|failed"
done
