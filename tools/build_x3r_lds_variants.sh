#!/bin/bash
# LDS bank-conflict attribution of conv_x3r_kernel: the probe harness WITHOUT the s_memtime stamps, one build per instruction class dropped
for spec in base:"" nostore:"-DXR_X_NOSTORE" noa:"-DXR_X_NOA" nored:"-DXR_X_NORED" noepi:"-DXR_X_NOEPI" none:"-DXR_X_NOSTORE -DXR_X_NOA -DXR_X_NORED -DXR_X_NOEPI"; do
  name=${spec%%:*}; flags=${spec#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags -Iinclude -Isatlas_super_resolution_amd/csrc tools/x3r_probe.hip -o tools/x3r_l_$name 2>/dev/null &
done
wait
ls -la tools/x3r_l_*
