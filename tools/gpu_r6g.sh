#!/bin/bash
cd /root/repo
python -X faulthandler - <<'PY' 2>&1 | tail -25
import os, sys, tempfile
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import test_dp_gpu as T
d = tempfile.mkdtemp()
T._rccl_worker(T._free_port(), d)
print("worker done", os.listdir(d))
PY
timeout 600 python -m pytest -q -p no:cacheprovider tests/test_dp_gpu.py -x -m gpu -k rccl_world1 2>&1 | tail -3
