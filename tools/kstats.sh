#!/bin/bash
# per-kernel time of one build of libssr_hip.so: tools/kstats.sh ab/x.so tag   -> gpurun_out/kstats_<tag>.csv (top kernels printed)
SO=$1; TAG=$2
L=satlas_super_resolution_amd/libssr_hip.so
R=$(pwd)
cp $L /tmp/keep.so; cp $SO $L
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_$TAG.log 2>&1)
F=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
[ -z "$F" ] && { tail -20 /tmp/prof_$TAG.log; find /tmp/prof_$TAG | head; }
cp $F $R/gpurun_out/kstats_$TAG.csv
python - "$F" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    n=re.sub(r'\(.*','',r['Name'])[:70]
    print(f"{n:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/13/1e3:9.1f} us/step  avg {float(r['AverageNs'])/1e3:8.1f} us")
print('total/step us', tot/13/1e3)
PY
cp /tmp/keep.so $L
