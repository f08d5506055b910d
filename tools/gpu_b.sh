#!/bin/bash
# second visit of round 2: bench with a phase trace (the first one hung past its CPU baseline), the failed / new tests, stats
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=r02b
timeout 420 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; tail -12 $O/${TAG}_bench.err; cut -c1-2500 $O/${TAG}_bench.json
timeout 600 python -m pytest -q -p no:cacheprovider -s tests/test_gpu_baseline_shapes.py tests/test_gpu_perceptual.py -k "bf16 or x3 or perceptual or maxpool" --durations=8 > $O/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; grep -E "^\[|passed|failed" $O/${TAG}_tests.log | cut -c1-420 | tail -40; grep -E "^(FAILED|ERROR)" $O/${TAG}_tests.log | head -30
grep -n "Error" $O/${TAG}_tests.log | cut -c1-700 | head -12
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/prof_$TAG.log 2>&1)
F=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $O/${TAG}_kernel_stats.csv && head -12 $F | cut -c1-150
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d /tmp/pmc_$TAG -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pmc_$TAG.log 2>&1)
python tools/pmc_sq.py /tmp/pmc_$TAG $O/${TAG}_pmc_sq.json || tail -20 /tmp/pmc_$TAG.log
