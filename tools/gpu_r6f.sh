#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest -q -p no:cacheprovider tests/test_dp_gpu.py -x -m gpu -k rccl_world1 2>&1 | grep -v "^$" | tail -60
python bench.py --no-cpu-baseline --no-legs > $O/r06f_bench.json 2> $O/r06f_bench.err; echo "bench rc=$?"; tail -3 $O/r06f_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06f_bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['rocprof_symbols'], round(d['roofline']['frac'],4), d['roofline']['avg_launch_us'], d['roofline']['launches_per_step'], d['roofline']['measured_on'])
print(d['kernel_time_breakdown_ms'])
PY
