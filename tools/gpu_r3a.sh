#!/bin/bash
# r03a: second-generation dense-block kernel vs the first: bit-identity + timing, then A/B bench lines
O=gpurun_out; mkdir -p $O
timeout 240 tools/rdbt_check all > $O/r03a_check.log 2>&1; echo "check rc=$?"; tail -45 $O/r03a_check.log
if grep -q "check: 0 failing" $O/r03a_check.log; then
  for t in 0 16 8; do
    SSR_RDB_TILE=$t timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --blocks-timed 2 > $O/r03a_bench_tile$t.json 2> $O/r03a_bench_tile$t.err; echo "tile $t rc=$?"; cut -c1-400 $O/r03a_bench_tile$t.json
  done
  SSR_RDB_TILE=16 SSR_G_SPLIT=0 timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --no-roofline --blocks-timed 2 > $O/r03a_bench_tile16_nosplit.json 2>&1; cut -c1-300 $O/r03a_bench_tile16_nosplit.json
fi
