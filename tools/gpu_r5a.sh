#!/bin/bash
# r05a: where the fp32x3 step's time is (per-layer dump of the instrumented step + serial rocprofv3 kernel stats), and the round's new
# multi-rank tests.
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
SSR_BENCH_LAYER_DUMP=$O/r05a_layers_fp32x3.txt timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 10 --warmup 3 --blocks-timed 1 > $O/r05a_bench_fp32x3.json 2> $O/r05a_bench.err; echo "bench rc=$?"
tail -1 $O/r05a_bench_fp32x3.json | cut -c1-600
(cd /tmp && SSR_OVERLAP_D=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs -o k -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-legs > /tmp/profs.log 2>&1)
F=$(find /tmp/profs -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $O/r05a_kernel_stats_serial_fp32x3.csv && head -14 $F | cut -c1-150
timeout 1500 python -m pytest -q -p no:cacheprovider tests/test_dp_gpu.py tests/test_gpu_infer_scripts.py tests/test_gpu_boundary.py -m gpu -x -s -k "n_ranks or shared_gpu or resume or test_pipeline" > $O/r05a_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|infer_grid," $O/r05a_tests.log | tail -6
