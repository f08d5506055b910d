#!/bin/bash
# round 6: the fp32h rows added to the layer-local / whole-tile tests, whole-tile inference end to end in the default arithmetic
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest -q -p no:cacheprovider -s tests/test_gpu_baseline_shapes.py -m gpu -k "every_layer and fp32h or infer_grid_tile" 2>&1 | grep -E "layerwise|infer tile|passed|failed|Error|assert" | cut -c1-400 | tee $O/r06n_fp32h_layerwise.txt
timeout 600 python tools/infer_e2e_bench.py fp32h 12 > $O/r06n_infer_e2e_fp32h.json 2> $O/r06n_infer_e2e_fp32h.err; tail -c 1500 $O/r06n_infer_e2e_fp32h.json
