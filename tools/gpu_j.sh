#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python -m pytest -q -p no:cacheprovider tests/test_gpu_boundary.py -k "training_loop or validation or shipped or resume" > $O/r02j_tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/r02j_tests.log | cut -c1-220; grep -n "Error" $O/r02j_tests.log | cut -c1-500 | head -6
timeout 200 python bench.py --no-cpu-baseline --no-parity-mode --blocks-timed 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'])"
SSR_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --blocks-timed 0 2>$O/r02j_dp2.err | cut -c1-600
tail -3 $O/r02j_dp2.err | cut -c1-300
