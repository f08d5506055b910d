#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest -q -p no:cacheprovider tests/test_abi.py tests/test_dp_gpu.py tests/test_gpu_conv_x3.py tests/test_gpu_parity.py -x -m gpu --durations=8 2>&1 | tail -16
python bench.py --no-cpu-baseline > $O/r06e_bench.json 2> $O/r06e_bench.err; echo "bench rc=$?"; tail -3 $O/r06e_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06e_bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['rocprof_symbols'], round(d['roofline']['frac'],4), d['roofline']['avg_launch_us'])
print(d.get('modes')); print(d.get('value_all_gates'))
print(d['kernel_time_breakdown_ms'])
PY
