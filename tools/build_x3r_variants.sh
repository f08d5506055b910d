#!/bin/bash
# builds tools/x3r_probe.hip as tools/x3r_x_<name> for every variant a GPU-box visit compares: tools/build_x3r_variants.sh name:"-Dflags" ...
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSR_PROBE $flags -Iinclude -Isatlas_super_resolution_amd/csrc tools/x3r_probe.hip -o tools/x3r_x_$name &
done
wait
ls -la tools/x3r_x_*
