#!/bin/bash
# round 6: the parity classes of the small-grid stride-2 dgrads on the big-tile kernel (half-empty 32-row tiles) against the pipelined kernel - A/B in one call
for r in 1 2; do
  for v in "SSR_X3_BIGTILE2=1" "SSR_X3_BIGTILE2=2"; do
    echo -n "$v  "; env $v python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
SSR_X3_BIGTILE2=2 python bench.py --no-cpu-baseline --no-legs --blocks-timed 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(json.dumps(d['kernel_time_breakdown_ms']))"
