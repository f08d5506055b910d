#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1700 python -m pytest -q -p no:cacheprovider tests -m gpu --durations=15 -x 2>&1 | tail -30 > $O/r06m_tests_tail.txt; tail -22 $O/r06m_tests_tail.txt
