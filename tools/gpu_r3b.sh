#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 240 tools/rdbt_check ${1:-all} > $O/r03b_check.log 2>&1; echo "check rc=$?"; grep -E "MISMATCH|check:|time|failed" $O/r03b_check.log | head -60
