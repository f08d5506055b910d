#!/bin/bash
# bench under several environment settings inside ONE gpurun call: tools/ab_envn.sh "VAR=a" "VAR=b" ... (two rounds)
for r in 1 2; do
  for v in "$@"; do
    echo -n "$v  "; env $v python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
