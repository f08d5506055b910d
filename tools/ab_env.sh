#!/bin/bash
# A/B of one build under two environments inside ONE gpurun call: tools/ab_env.sh "VAR=a" "VAR=b" [bench args]
A=$1; B=$2; shift 2
for r in 1 2; do
  for v in "$A" "$B"; do
    echo "== $v"; env $v python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
