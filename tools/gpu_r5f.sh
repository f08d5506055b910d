#!/bin/bash
# r05f: the whole -m gpu suite on the round's build
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest -q -p no:cacheprovider tests -m gpu --durations=15 -s > $O/r05f_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|rc=" $O/r05f_tests.log | tail -5; grep -E "^(FAILED|ERROR)" $O/r05f_tests.log | head -30
