#!/bin/bash
# r03f: step A/B with the second-generation dense-block kernel
O=gpurun_out; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --blocks-timed 2 > $O/r03f_$tag.json 2> $O/r03f_$tag.err; echo "$tag rc=$? $(python -c "import json,sys; d=json.loads(open('$O/r03f_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'))" 2>&1 | tail -1)"; }
run old SSR_RDB_TILE=0
run new_nosplit SSR_RDB_TILE=16 SSR_G_SPLIT=0
run new_split SSR_RDB_TILE=16 SSR_G_SPLIT=2
run old_nosplit SSR_RDB_TILE=0 SSR_G_SPLIT=0
run new_b16 SSR_RDB_TILE=16 BENCH_ARGS=x
