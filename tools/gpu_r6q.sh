#!/bin/bash
timeout 1500 python -m pytest -q -p no:cacheprovider tests/test_gpu_baseline_shapes.py -m gpu -k "train_step_full_depth and (fp32h or fp32x3) or discriminator_every_layer and fp32x3" 2>&1 | tail -6
