#!/bin/bash
# the whole -m gpu suite at the driver's depth; failures listed
O=gpurun_out; mkdir -p $O
timeout 1700 python -m pytest -q -p no:cacheprovider tests -m gpu --durations=15 2>&1 | tail -60 > $O/r06t_tests_tail.txt; grep -E "^(FAILED|ERROR)|passed|failed" $O/r06t_tests_tail.txt | tail -30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
