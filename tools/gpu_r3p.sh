#!/bin/bash
for a in "69 2 1 1" "8 2 0 1" "16 2 0 1" "32 2 0 1" "69 1 0 1" "69 0 0 1"; do timeout 60 tools/wgrad_body_probe $a | head -2; done
