#!/bin/bash
for r in 1 2; do for v in 0 1; do echo -n "SSR_X3_REGTILE_SPLIT=$v  "; SSR_X3_REGTILE_SPLIT=$v python bench.py --frames 1 --batch 16 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; done; done
timeout 600 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py -x -m gpu -k "regtile" 2>&1 | tail -2
