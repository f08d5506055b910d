#!/bin/bash
# A/B of an environment switch inside ONE gpurun call: tools/ab_env.sh VAR a b [bench args]
V=$1; A=$2; B=$3; shift 3
for r in 1 2; do
  for x in $A $B; do
    echo "== $V=$x"; env $V=$x python bench.py --steps 30 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
