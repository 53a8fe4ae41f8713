#!/bin/bash
# usage: [SRC="gemm attention"] tools/build_variant.sh NAME [-DFLAG ...]   -> vtc_amd/lib/variants/libvtc_NAME.so (tuning experiments;
# select with VTC_HIP_LIB=..., or load side by side: tools/gemm_ab.py).  Only the sources named in SRC (default: gemm) are
# recompiled with the extra flags; the other objects are the product build's.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
SRC=${SRC:-gemm}
make -s -C vtc_amd/csrc -j8
mkdir -p build/var_$name vtc_amd/lib/variants
objs=""
for f in gemm norm attention embed sweep towers prof train cam; do
  if [[ " $SRC " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-variable "$@" -c vtc_amd/csrc/$f.hip -o build/var_$name/$f.o
    objs="$objs build/var_$name/$f.o"
  else
    objs="$objs build/obj/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vtc_amd/lib/variants/libvtc_$name.so $objs
echo built vtc_amd/lib/variants/libvtc_$name.so
