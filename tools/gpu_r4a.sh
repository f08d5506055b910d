#!/bin/bash
# round 4, visit a: lesson 36 closed? (library variants x guard + stress tests; harness variants), decisive probes of the dense block
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=satlas_super_resolution_amd/libssr_hip.so
PYT="python -m pytest -q -p no:cacheprovider -m gpu -x"
cp $L /tmp/keep.so
{
echo "### library variants: guard test + stress test (SSR_STRESS_LAUNCHES=1500)"
for v in ab/repro_sel2_oldpoll.so ab/sel2_newpoll.so ab/sel4_newpoll.so ab/sel0_newpoll.so /tmp/keep.so; do
  cp $v $L
  echo "== $v"
  SSR_STRESS_LAUNCHES=1500 timeout 600 $PYT tests/test_gpu_rdb_tile.py tests/test_gpu_rdb_stress.py 2>&1 | grep -E "passed|failed|Error|differing" | head -5
done
cp /tmp/keep.so $L
echo "### harness: byte identity per select form + launch time"
for v in sel0 sel1 sel2 sel3 sel4; do echo "== $v"; timeout 120 tools/rdbt_x_$v check | tail -1; timeout 60 tools/rdbt_x_$v time32 | grep "tile 16"; done
echo "### harness: launch time of variants (B=32)"
for v in rq2 rq4 sel0rq prio10 prio00 sleep1 wc4 wc8; do echo "== $v"; timeout 60 tools/rdbt_x_$v time32 | grep "tile 16"; done
echo "### harness: 64 / 128 / 256 workgroups (per-block time vs number of busy CUs)"
timeout 120 tools/rdbt_x_sel1 timen 8 16 32 | grep "tile 16"
echo "### harness: chain of 69"
timeout 120 tools/rdbt_x_sel1 chain | tail -3
timeout 120 tools/rdbt_x_sel0 chain | tail -3
echo "### probes"
for v in p_sel0 p_sel1 p_nosync p_noprod p_wc4; do echo "== $v"; timeout 60 tools/rdbt_x_$v probe | grep -A14 "tile 16"; done
echo "### l2_probe2"
timeout 120 tools/l2_probe2
} > $O/r04a_probes.log 2>&1
python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('step', d['ms_per_step'], d['value'], r['kernel'], round(r['frac'],4), round(r['avg_launch_us'],2))" > $O/r04a_bench.txt 2>&1
tail -5 $O/r04a_bench.txt
grep -E "^==|passed|failed|identical|failing" $O/r04a_probes.log | head -60
