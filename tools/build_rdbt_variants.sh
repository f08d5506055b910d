#!/bin/bash
# Builds the stand-alone dense-block harness (tools/rdbt_check.hip) in the variants a GPU-box visit compares.
#   tools/build_rdbt_variants.sh name:"-Dflags" ...     -> tools/rdbt_x_<name>   (git-ignored; travel with the gpurun snapshot)
cd "$(dirname "$0")/.."
build() {
    name=${1%%:*}; flags=${1#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w $flags -Iinclude -Isatlas_super_resolution_amd/csrc \
        tools/rdbt_check.hip -o tools/rdbt_x_$name 2>&1 | grep -i "error" | head -5
    echo "built tools/rdbt_x_$name ($flags)"
}
for v in "$@"; do build "$v" & 
  while [ $(jobs -r | wc -l) -ge ${JOBS:-6} ]; do sleep 1; done
done
wait
