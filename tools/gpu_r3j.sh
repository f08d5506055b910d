#!/bin/bash
O=gpurun_out; mkdir -p $O
for m in fp32x3 bf16; do timeout 600 python tools/infer_e2e_bench.py $m 4 > $O/r03j_infer_e2e_$m.json 2> $O/r03j_infer_e2e_$m.err; echo "$m rc=$?"; python -c "
import json; d=json.loads(open('$O/r03j_infer_e2e_$m.json').read().strip().splitlines()[-1]); print(d['end_to_end'], d['generator_only'])"; tail -2 $O/r03j_infer_e2e_$m.err | grep -v amdgpu.ids; done
timeout 900 python -m pytest -q -p no:cacheprovider -m gpu -s tests/test_gpu_infer_scripts.py > $O/r03j_tests.log 2>&1; echo "pytest rc=$?"; grep -E "^\[|passed|failed|Error" $O/r03j_tests.log | cut -c1-200
