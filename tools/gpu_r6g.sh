#!/bin/bash
# round 6, last visit: the whole -m gpu suite on the final test files, then the default bench line (with roofline_by_kernel)
O=gpurun_out; mkdir -p $O
timeout 1700 python -m pytest -q -p no:cacheprovider tests -m gpu --durations=12 2>&1 | tail -40 > $O/r06g_tests_tail.txt; grep -E "^(FAILED|ERROR)|passed|failed" $O/r06g_tests_tail.txt | tail -10
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r06g_bench.json 2> $O/r06g_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r06g_bench.json').read().strip().splitlines()[-1])
print(d['dtype'], round(d['ms_per_step'],3), round(d['value'],1), d['roofline']['kernel'], round(d['roofline']['frac'],4), d['roofline']['traffic'], d['value_all_gates']['mode'])
for k,v in d['roofline_by_kernel'].items(): print('  ', k, v)
P
