#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -m gpu -s -k "vs_reference_class" > $O/r03h_tests.log 2>&1; echo "pytest rc=$?"; grep -E "masked|passed|failed|Error|error" $O/r03h_tests.log | cut -c1-260 | head -30
timeout 600 python bench.py > $O/r03h_bench.json 2> $O/r03h_bench.err; echo "bench rc=$?"; tail -3 $O/r03h_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03h_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], round(d['roofline']['frac'],4))
for k,v in d.get('legs',{}).items(): print(k, round(v['ms_per_step'],2), round(v['value'],1), v['roofline']['kernel'] if v['roofline'] else None, round(v['roofline']['frac'],4) if v['roofline'] else None, v['max_rel_err_vs_oracle'])
PY
