#!/bin/bash
# Build a second copy of the library with extra defines, for A/B runs of two builds on one box:
#   tools/build_variant.sh b -DSALNMF_LOGTAB_REP1   ->  salamander_amd/lib/libsalnmf_b.so   (run with SALNMF_LIB=<that path>)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
obj=build/obj_$tag; mkdir -p $obj salamander_amd/lib
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 $*"
c=salamander_amd/csrc
( /opt/rocm/bin/hipcc $flags -c $c/salnmf.hip -o $obj/salnmf.o ) &
( /opt/rocm/bin/hipcc $flags -c $c/salnmf_forward_inst.hip -o $obj/forward_inst.o ) &
( /opt/rocm/bin/hipcc $flags -c $c/salnmf_corr_batched.hip -o $obj/corr_batched.o ) &
( /opt/rocm/bin/hipcc $flags -c $c/salnmf_small.hip -o $obj/small.o ) &
wait
for i in 0 1 2 3 4 5; do ( /opt/rocm/bin/hipcc $flags -DSALNMF_GEOM_SET=$i -c $c/salnmf_fused_inst.hip -o $obj/fused_inst_$i.o ) & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o salamander_amd/lib/libsalnmf_$tag.so $obj/*.o -ldl
echo built salamander_amd/lib/libsalnmf_$tag.so
