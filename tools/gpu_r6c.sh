#!/bin/bash
bash tools/ab_envn.sh "SSR_G_SPLIT=auto" "SSR_G_SPLIT=2" "SSR_OVERLAP_D=0" 2>&1 | tail -8
