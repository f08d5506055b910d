#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 500 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py tests/test_gpu_perceptual.py tests/test_gpu_boundary.py -k "fp32x3 or shipped_option" > $O/r02n_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/r02n_tests.log | cut -c1-220; grep -n "Error" $O/r02n_tests.log | cut -c1-500 | head -6
(cd /tmp && timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pf.log 2>&1)
(cd /tmp && timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pw.log 2>&1)
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $O/r02n_traffic.json | tail -2
cp $O/r02n_traffic.json profiles/traffic.json
timeout 300 python bench.py > $O/r02n_bench.json 2> $O/r02n_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/r02n_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['parity_mode']['ms_per_step'], d['parity_mode']['value'], d['parity_mode']['max_rel_err_vs_oracle'], d['cpu_baseline']['value'])"
