#!/bin/bash
# after a comment-only change of a hashed source: dense-block guard tests, PMC traffic passes, bench line
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest -q -p no:cacheprovider -m gpu -x tests/test_gpu_rdb_tile.py tests/test_gpu_parity.py 2>&1 | tail -1
TAG=r03z
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch_$TAG -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pmcf_$TAG.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write_$TAG -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode > /tmp/pmcw_$TAG.log 2>&1)
python tools/pmc_traffic.py /tmp/pmc_fetch_$TAG /tmp/pmc_write_$TAG $O/${TAG}_traffic.json > /dev/null && cp $O/${TAG}_traffic.json profiles/traffic.json && python -c "
import json; d=json.load(open('$O/${TAG}_traffic.json')); print(d['_meta']['source_hash'][:12]); [print(k, round(v['hbm_read_bytes_per_launch']/1e6,1), 'MB read', round(v['hbm_write_bytes_per_launch']/1e6,1), 'MB written') for k,v in d.items() if k!='_meta' and ('rdbt' in k or 'wgrad' in k)]"
python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('step', d['ms_per_step'], d['value'], r['kernel'], round(r['frac'],4), round(r['avg_launch_us'],2), 'traffic', r['traffic'])"
