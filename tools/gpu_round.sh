#!/bin/bash
# One GPU-box visit: the whole -m gpu suite, the default bench line, rocprofv3 kernel stats, SQ/GRBM counters.
#   gpurun --timeout 1700 -- 'bash tools/gpu_round.sh r02a'
TAG=${1:-run}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
PYT="python -m pytest -q -p no:cacheprovider"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1100 $PYT tests -m gpu --durations=12 -s ${PYTEST_ARGS:-} > $O/${TAG}_tests.log 2>&1
  echo "pytest rc=$?" >> $O/${TAG}_tests.log
  grep -E "^\[|passed|failed|rc=" $O/${TAG}_tests.log | tail -60
  grep -E "^(FAILED|ERROR)" $O/${TAG}_tests.log | head -40
fi
python bench.py ${BENCH_ARGS:-} > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; cat $O/${TAG}_bench.json | cut -c1-1500
if [ "${SKIP_PROF:-0}" != "1" ]; then
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/prof_$TAG.log 2>&1)
  F=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $O/${TAG}_kernel_stats.csv && head -14 $F | cut -c1-160
  (cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY \
      --kernel-trace --output-format csv -d /tmp/pmc_$TAG -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pmc_$TAG.log 2>&1)
  python tools/pmc_sq.py /tmp/pmc_$TAG $O/${TAG}_pmc_sq.json || tail -20 /tmp/pmc_$TAG.log
fi
if [ "${LEG_TRAFFIC:-0}" = "1" ]; then      # HBM traffic of the parity legs' dominant kernels -> profiles/traffic_<dtype>.json (bench.py legs.*.roofline.traffic)
  for DT in fp32x3 fp32; do
    (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch_${TAG}_$DT -- python $R/bench.py --dtype $DT --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pmcf_${TAG}_$DT.log 2>&1)
    (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write_${TAG}_$DT -- python $R/bench.py --dtype $DT --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pmcw_${TAG}_$DT.log 2>&1)
    SSR_PMC_DTYPE=$DT python tools/pmc_traffic.py /tmp/pmc_fetch_${TAG}_$DT /tmp/pmc_write_${TAG}_$DT $O/${TAG}_traffic_$DT.json > /dev/null && cp $O/${TAG}_traffic_$DT.json profiles/traffic_$DT.json && echo "traffic $DT ok"
  done
fi
if [ "${SKIP_TRAFFIC:-0}" != "1" ]; then
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch_$TAG -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pmcf_$TAG.log 2>&1)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write_$TAG -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pmcw_$TAG.log 2>&1)
  python tools/pmc_traffic.py /tmp/pmc_fetch_$TAG /tmp/pmc_write_$TAG $O/${TAG}_traffic.json > /dev/null && python -c "
import json; d=json.load(open('$O/${TAG}_traffic.json')); [print(k, round(v['hbm_read_bytes_per_launch']/1e6,1), 'MB read', round(v['hbm_write_bytes_per_launch']/1e6,1), 'MB written') for k,v in d.items() if k!='_meta']"
fi
