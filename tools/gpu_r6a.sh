#!/bin/bash
# round 6, visit a: the register-tiled body kernel (csrc/conv_x3r.hip) - probes beside the ring kernel, its parity + stress tests, A/B of the step
O=gpurun_out; mkdir -p $O
{
for a in "32 64 32" "32 96 32" "32 128 32" "32 160 32" "32 192 64" "32 64 32 32 32 2" "32 192 64 32 32 1" "16 128 32" "16 192 64"; do timeout 60 tools/x3r_probe $a; done
} > $O/r06a_x3r_probe.txt 2>&1
tail -n +1 $O/r06a_x3r_probe.txt | grep -E "avg launch|whole|sum \+|first chunk" 
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_gpu_conv_x3.py tests/test_gpu_x3_stress.py -x -m gpu 2>&1 | tail -8
bash tools/ab_env.sh SSR_X3_REGTILE 0 1 --no-legs --no-roofline 2>&1 | tail -8
