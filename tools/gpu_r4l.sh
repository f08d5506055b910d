#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest -q -p no:cacheprovider -m gpu "tests/test_gpu_infer_scripts.py" -x 2>&1 | grep -E "^E |Error|error|passed|failed" | head -30
