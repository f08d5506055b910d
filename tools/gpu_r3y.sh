#!/bin/bash
# python-level knobs of the weight-gradient launch on the final build, one box
run() { env "$@" python bench.py --no-parity-mode --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', round(d['ms_per_step'],3), round(d['value'],1))"; }
run A=0
run SSR_WGRAD_T3=64
run SSR_WGRAD_T3=256
run SSR_WGRAD_BALANCE=1
run SSR_WGRAD_T4=32
run SSR_WGRAD_T4=128
run A=1
