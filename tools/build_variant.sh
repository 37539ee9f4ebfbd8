#!/bin/bash
# Build an experimental variant of libjxlamd.so: one translation unit recompiled with extra -D flags, the rest reused from
# jxl_coder_amd/build/.  Usage: tools/build_variant.sh <name> <tu.hip> "<flags>"   ->  jxl_coder_amd/libjxlamd_<name>.so
# Run a tool against it with JXLAMD_LIB=jxl_coder_amd/libjxlamd_<name>.so.
set -e
cd "$(dirname "$0")/../jxl_coder_amd"
name=$1; tu=$2; flags=$3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c csrc/$tu -o build/$tu.$name.o
objs=""
for o in build/*.o; do
  case "$o" in *.$name.o|build/$tu.o) ;; *.*.*.o) ;; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o libjxlamd_$name.so $objs build/$tu.$name.o
echo built libjxlamd_$name.so
