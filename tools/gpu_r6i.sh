#!/bin/bash
bash tools/ab_envn.sh "SSR_X3_CHAIN=0" "SSR_X3_CHAIN=1" 2>&1 | tail -4
