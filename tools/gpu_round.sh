#!/bin/bash
# One GPU-box visit: the whole -m gpu suite, HBM-traffic PMC passes (the headline mode - HEAD_DTYPE, default fp32h - + the bf16 / fp32 legs), the default bench
# line (which then carries roofline.traffic for this very build), rocprofv3 kernel stats of the headline mode (overlapped and serial),
# SQ/GRBM counters, and the other BASELINE.json configurations.
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r05z'
# Env: SKIP_TESTS=1, SKIP_TRAFFIC=1, SKIP_LEG_TRAFFIC=1, SKIP_PROF=1, SKIP_CONFIGS=1, PYTEST_ARGS, BENCH_ARGS
TAG=${1:-run}
HEAD=${HEAD_DTYPE:-fp32h}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
PYT="python -m pytest -q -p no:cacheprovider"
PMC_BENCH="--steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-legs"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  # the driver's run: default depth (slow variants skipped); then the slow variants and the deep stress loops on their own
  timeout 1800 $PYT tests -m gpu --durations=12 -s ${PYTEST_ARGS:-} > $O/${TAG}_tests.log 2>&1
  echo "pytest rc=$?" >> $O/${TAG}_tests.log
  grep -E "passed|failed|rc=" $O/${TAG}_tests.log | tail -5
  grep -E "^(FAILED|ERROR)" $O/${TAG}_tests.log | head -40
  if [ "${SKIP_SLOW:-0}" != "1" ]; then
    SSR_RUN_SLOW=1 timeout 900 $PYT tests -m "gpu and slow" -s > $O/${TAG}_tests_slow.log 2>&1
    SSR_STRESS_LAUNCHES=2500 SSR_STRESS_LAUNCHES_X3=4000 timeout 900 $PYT tests/test_gpu_rdb_stress.py tests/test_gpu_x3_stress.py -m gpu -s >> $O/${TAG}_tests_slow.log 2>&1
    echo "pytest (slow + deep stress) rc=$?" >> $O/${TAG}_tests_slow.log
    grep -E "passed|failed|rc=" $O/${TAG}_tests_slow.log | tail -3
  fi
fi
if [ "${SKIP_TRAFFIC:-0}" != "1" ]; then      # separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one), --kernel-trace only
  # (SSR_G_SPLIT=0: full-batch launches, the launch shape roofline.kernel is measured at - bench.py unsplit_twin - and the serial trace uses)
  (cd /tmp && SSR_G_SPLIT=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch_$TAG -- python $R/bench.py --dtype $HEAD $PMC_BENCH > /tmp/pmcf_$TAG.log 2>&1)
  (cd /tmp && SSR_G_SPLIT=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write_$TAG -- python $R/bench.py --dtype $HEAD $PMC_BENCH > /tmp/pmcw_$TAG.log 2>&1)
  SSR_PMC_DTYPE=$HEAD python tools/pmc_traffic.py /tmp/pmc_fetch_$TAG /tmp/pmc_write_$TAG $O/${TAG}_traffic_$HEAD.json > /dev/null && cp $O/${TAG}_traffic_$HEAD.json profiles/traffic_$HEAD.json && python -c "
import json; d=json.load(open('$O/${TAG}_traffic_$HEAD.json')); [print(k, round(v['hbm_read_bytes_per_launch']/1e6,1), 'MB read', round(v['hbm_write_bytes_per_launch']/1e6,1), 'MB written') for k,v in d.items() if k!='_meta']"
fi
if [ "${SKIP_LEG_TRAFFIC:-0}" != "1" ]; then  # the parity legs' dominant kernels -> profiles/traffic_<dtype>.json (bench.py legs.*.roofline.traffic)
  for DT in bf16 fp32; do
    (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch_${TAG}_$DT -- python $R/bench.py --dtype $DT ${PMC_BENCH/--steps 2/--steps 1} > /tmp/pmcf_${TAG}_$DT.log 2>&1)
    (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write_${TAG}_$DT -- python $R/bench.py --dtype $DT ${PMC_BENCH/--steps 2/--steps 1} > /tmp/pmcw_${TAG}_$DT.log 2>&1)
    TJ=profiles/traffic_$DT.json; [ "$DT" = bf16 ] && TJ=profiles/traffic.json
    SSR_PMC_DTYPE=$DT python tools/pmc_traffic.py /tmp/pmc_fetch_${TAG}_$DT /tmp/pmc_write_${TAG}_$DT $O/${TAG}_traffic_$DT.json > /dev/null && cp $O/${TAG}_traffic_$DT.json $TJ && echo "traffic $DT collected"
  done
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py ${BENCH_ARGS:-} > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; tail -1 $O/${TAG}_bench.json | cut -c1-1200
if [ "${SKIP_PROF:-0}" != "1" ]; then
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-legs > /tmp/prof_$TAG.log 2>&1)
  F=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $O/${TAG}_kernel_stats.csv && head -8 $F | cut -c1-160
  (cd /tmp && SSR_OVERLAP_D=0 SSR_G_SPLIT=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$TAG -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-legs > /tmp/profs_$TAG.log 2>&1)
  F=$(find /tmp/profs_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $O/${TAG}_kernel_stats_serial.csv
  (cd /tmp && SSR_G_SPLIT=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY \
      --kernel-trace --output-format csv -d /tmp/pmc_$TAG -- python $R/bench.py $PMC_BENCH > /tmp/pmc_$TAG.log 2>&1)
  python tools/pmc_sq.py /tmp/pmc_$TAG $O/${TAG}_pmc_sq.json > /dev/null || tail -20 /tmp/pmc_$TAG.log
fi
# roofline.frac recomputed from the serial kernel trace by name (tools/roofline_check.py): profiles/<tag>_roofline_check.txt
[ -f $O/${TAG}_kernel_stats_serial.csv ] && python tools/roofline_check.py $O/${TAG}_bench.json $O/${TAG}_kernel_stats_serial.csv > $O/${TAG}_roofline_check.txt 2>&1 && head -12 $O/${TAG}_roofline_check.txt
if [ "${SKIP_DP:-0}" != "1" ]; then           # the data-parallel branch over RCCL at world size 1: per-rank diagnostics of both launch forms
  : > $O/${TAG}_dp_world1.jsonl
  for OG in 0 1; do for ALGO in allreduce rsag; do
    SSR_DP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + OG * 2 + ${#ALGO})) SSR_DP_ONE_GRAPH=$OG SSR_DP_ALGO=$ALGO \
      python bench.py --no-cpu-baseline --no-legs --no-roofline --blocks-timed 2 2>/dev/null | tail -1 >> $O/${TAG}_dp_world1.jsonl
  done; done
  python -c "
import json
for ln in open('$O/${TAG}_dp_world1.jsonl'):
    d = json.loads(ln); p = d['dp']; print('one_graph', p['one_graph'], p['algo'], round(d['ms_per_step'], 3), 'ms/step; host enqueue', p['per_rank_host_enqueue_ms_per_step'], 'comm busy', p['per_rank_comm_stream_busy_ms_per_step'])"
fi
if [ "${SKIP_CONFIGS:-0}" != "1" ]; then      # the other BASELINE.json configurations on this build, one JSON line each
  : > $O/${TAG}_bench_configs.jsonl
  for ARGS in "--frames 1 --batch 16" "--frames 32 --batch 16" "--feed-disc-lr" "--perceptual" "--dtype fp32x3" "--dtype fp32x3 --frames 1 --batch 16" "--dtype bf16" "--dtype bf16 --frames 1 --batch 16" "--dtype bf16 --frames 32 --batch 16" "--dtype fp32f" "--dtype fp32"; do
    SSR_VGG19_RANDOM=1 python bench.py $ARGS --no-cpu-baseline --no-legs 2>/dev/null | tail -1 >> $O/${TAG}_bench_configs.jsonl
  done
  SSR_DETERMINISTIC=1 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); d['config']['deterministic'] = True; print(json.dumps(d))" >> $O/${TAG}_bench_configs.jsonl
  python -c "
import json
for ln in open('$O/${TAG}_bench_configs.jsonl'):
    d = json.loads(ln); print(round(d['ms_per_step'], 3), 'ms', round(d['value'], 1), 'img/s', d['dtype'], d['config'].get('workload', '')[:70], 'det' if d['config'].get('deterministic') else '')"
fi
