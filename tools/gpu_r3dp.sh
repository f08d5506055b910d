#!/bin/bash
# the data-parallel branch (3 backward segments, RCCL exchanges on the comm stream) at world size 1 against the single-process step, same box
run() { env "$@" MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --no-parity-mode --no-cpu-baseline --no-roofline 2>/tmp/err.log | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('parallelism'))" || tail -3 /tmp/err.log; }
run A=0
run SSR_DP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run SSR_DP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSR_DP_SEGMENTS=1
run A=1
