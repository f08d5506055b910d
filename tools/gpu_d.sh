#!/bin/bash
# full -m gpu suite + smoke + fp32-exact bench + PMC traffic passes for this build
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=r02d
timeout 900 python -m pytest -q -p no:cacheprovider tests -m gpu --durations=10 > $O/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; tail -18 $O/${TAG}_tests.log | cut -c1-260; grep -n "Error" $O/${TAG}_tests.log | cut -c1-600 | head -12
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 200 python bench.py --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --blocks-timed 0 > $O/${TAG}_bench_fp32_cfg2.json 2>/dev/null; cut -c1-700 $O/${TAG}_bench_fp32_cfg2.json
(cd /tmp && timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pf.log 2>&1)
(cd /tmp && timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pw.log 2>&1)
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $O/${TAG}_traffic.json | tail -3
