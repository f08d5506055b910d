#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
P='import sys,json; L=[l for l in sys.stdin.read().splitlines() if l.startswith("{")]; d=json.loads(L[-1]); print(len(L), d["n_gpus"], d["ms_per_step"], d["value"], d["losses_finite"])'
echo "== RCCL world 1 forced, B=32"; for f in 1 0; do SSR_DP_FORK=$f SSR_DP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2954$f timeout 200 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --blocks-timed 0 --no-parity-mode 2>$O/r02q_w1_$f.err | python -c "$P" || tail -3 $O/r02q_w1_$f.err; done
for f in 1 0; do
echo "== 2 ranks on one GPU (gloo), B=32, SSR_DP_FORK=$f"; SSR_DP_FORK=$f SSR_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2953$f bench.py --gpus 2 --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --blocks-timed 0 2>$O/r02q_w2_$f.err | python -c "$P" || tail -3 $O/r02q_w2_$f.err
done
