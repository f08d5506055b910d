#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest -q -p no:cacheprovider -m gpu tests/test_gpu_infer_scripts.py tests/test_gpu_baseline_shapes.py -k "infer or tile" 2>&1 | tail -3
for m in fp32x3 bf16; do timeout 600 python tools/infer_e2e_bench.py $m 12 2> $O/r04j_infer_e2e_$m.err | tail -1 > $O/r04j_infer_e2e_$m.json; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/r04j_infer_e2e_$m.json')); print('$m', d['end_to_end'], d['end_to_end_threads'], d['generator_only']['tiles_per_s'], d['io_workers'], d['host_cores'], d['pool_startup_s'])"; done
