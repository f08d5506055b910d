#!/bin/bash
# round 6: HBM read traffic of the one-pass weight-gradient launch under the two item orders (longest-first / XCD-grouped), and the step time beside it
O=$PWD/gpurun_out; mkdir -p $O; R=$PWD
export TMPDIR=/tmp
PMC_BENCH="--steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-legs"
for ORD in heavy hybrid; do
  (cd /tmp && SSR_WGRAD_ORDER=$ORD SSR_G_SPLIT=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$ORD -- python $R/bench.py $PMC_BENCH > /tmp/pmcf_$ORD.log 2>&1)
  python - <<P
import csv, glob
tot, n = {}, {}
for f in glob.glob('/tmp/pmc_f_$ORD/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE' or 'wgrad_x3' not in r['Kernel_Name']: continue
        k = 'k4' if 'k4' in r['Kernel_Name'] else 'k3'
        tot[k] = tot.get(k, 0.0) + float(r['Counter_Value']); n[k] = n.get(k, 0) + 1
for k in sorted(tot): print('order $ORD', 'wgrad_x3_' + k, 'launches', n[k], 'HBM read per launch (FETCH_SIZE x 2 KiB):', round(2 * tot[k] / n[k] * 1024 / 1e9, 3), 'GB')
P
done 2>&1 | tee $O/r06x3_wgrad_hybrid_order.txt
for r in 1 2 3; do for ORD in heavy hybrid; do echo -n "SSR_WGRAD_ORDER=$ORD  "; SSR_WGRAD_ORDER=$ORD python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"; done; done 2>&1 | tee -a $O/r06x3_wgrad_hybrid_order.txt
