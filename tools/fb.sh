#!/bin/bash
# build + run tools/fused_bench.hip against a given header directory (default: the product's csrc)
DIR=${1:-salamander_amd/csrc}; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I$DIR tools/fused_bench.hip -o /tmp/fb_$$ "$@" && /tmp/fb_$$
