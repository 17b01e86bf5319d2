#!/bin/bash
# usage: tools/build_abl.sh <name> "<-D flags>" <file.hip> [file.hip ...]
# Builds tools/abl/lib_<name>.so = the in-tree library with the listed sources recompiled under extra -D flags
# (ablation / tuning builds; select one at run time with L4D_LIB=tools/abl/lib_<name>.so).
set -e
cd "$(dirname "$0")/../lidar4d_amd/csrc"
name=$1; defs=$2; shift 2
make -s -j8 >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function"
mkdir -p ../../tools/abl/obj_$name
objs=""
for src in hashgrid planes mlp render optim fused field_bwd binscatter chamfer convert glue; do
  if [[ " $* " == *" $src.hip "* ]]; then
    extra=""; [[ $src == mlp && "$defs" != *NO_VGPR_FORM* ]] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    /opt/rocm/bin/hipcc $FLAGS $extra $defs -c $src.hip -o ../../tools/abl/obj_$name/$src.o
    objs="$objs ../../tools/abl/obj_$name/$src.o"
  else
    objs="$objs $src.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs capi.o -o ../../tools/abl/lib_$name.so
rm -rf ../../tools/abl/obj_$name
echo "built tools/abl/lib_$name.so ($defs)"
