#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1700 python -m pytest -q -p no:cacheprovider tests -m gpu --durations=12 2>&1 | tail -24 > $O/r06t_tests_tail.txt; tail -20 $O/r06t_tests_tail.txt
