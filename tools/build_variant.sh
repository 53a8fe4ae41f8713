#!/bin/bash
# usage: tools/build_variant.sh NAME [-DFLAG ...]   -> vtc_amd/lib/variants/libvtc_NAME.so (tuning experiments;
# select with VTC_HIP_LIB=...).  Only gemm.hip is recompiled with the extra flags.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
make -s -C vtc_amd/csrc -j8
mkdir -p build/var_$name vtc_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-variable "$@" -c vtc_amd/csrc/gemm.hip -o build/var_$name/gemm.o
objs=""
for f in qkv_attn norm attention embed sweep towers prof train cam; do objs="$objs build/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vtc_amd/lib/variants/libvtc_$name.so build/var_$name/gemm.o $objs
echo built vtc_amd/lib/variants/libvtc_$name.so
