#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1000 python -m pytest -q -p no:cacheprovider tests -m gpu --durations=5 > $O/r02l_tests.log 2>&1
echo "pytest rc=$?"; tail -10 $O/r02l_tests.log | cut -c1-220; grep -n "Error" $O/r02l_tests.log | cut -c1-600 | head -10
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python bench.py > $O/r02l_bench.json 2> $O/r02l_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/r02l_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('full_batch_launch'), d['parity_mode']['ms_per_step'], d['parity_mode']['max_rel_err_vs_oracle'], d['cpu_baseline']['value'])"
