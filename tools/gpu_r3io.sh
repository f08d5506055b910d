#!/bin/bash
O=gpurun_out; mkdir -p $O
for m in bf16 fp32x3; do timeout 500 python tools/infer_e2e_bench.py $m 12 > $O/r03io_infer_e2e_$m.json 2> $O/r03io_infer_e2e_$m.err; echo "$m rc=$?"; python -c "
import json; d=json.loads(open('$O/r03io_infer_e2e_$m.json').read().strip().splitlines()[-1]); print(d['end_to_end'], d['end_to_end_threads'], d['generator_only']['tiles_per_s'], d['host_cores'])"; tail -2 $O/r03io_infer_e2e_$m.err | grep -v amdgpu.ids; done
timeout 600 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_infer_scripts.py 2>&1 | tail -1
