#!/bin/bash
# end-of-round visit: the whole -m gpu suite, smoke, the default bench line, other configs, kernel stats, SQ counters, PMC traffic
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TAG=${1:-r02h}
timeout 1000 python -m pytest -q -p no:cacheprovider tests -m gpu --durations=6 > $O/${TAG}_tests.log 2>&1
echo "pytest rc=$?"; tail -12 $O/${TAG}_tests.log | cut -c1-220; grep -n "Error" $O/${TAG}_tests.log | cut -c1-500 | head -10
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-400 $O/${TAG}_bench.json
for cfg in "--frames 1 --batch 16" "--frames 32 --batch 16" "--feed-disc-lr"; do
  timeout 200 python bench.py $cfg --no-cpu-baseline --no-parity-mode --blocks-timed 3 2>/dev/null | tee -a $O/${TAG}_bench_other.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['ms_per_step'], d['value'])"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/prof_$TAG.log 2>&1)
F=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $O/${TAG}_kernel_stats.csv && head -8 $F | cut -c1-150
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d /tmp/pmc_$TAG -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pmc_$TAG.log 2>&1)
python tools/pmc_sq.py /tmp/pmc_$TAG $O/${TAG}_pmc_sq.json | grep -E "rdb|wgrad_bf16_k3" 
(cd /tmp && timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pf.log 2>&1)
(cd /tmp && timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pw.log 2>&1)
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $O/${TAG}_traffic.json | tail -3
