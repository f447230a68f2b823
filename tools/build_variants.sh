#!/bin/bash
# Timing-experiment builds of one source file: tools/build_variants.sh conv "-DDPMN_IGEMM_ABLATE=1" v1  ->  tools/variants/libdpmn_v1.so
# (the other objects come from the normal build; select with DPMN_HIP_LIB=tools/variants/libdpmn_v1.so)
set -e
cd "$(dirname "$0")/../dpmn_amd/csrc"
make -s
src=$1; flags=$2; tag=$3
mkdir -p ../../tools/variants build/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops $flags -c $src.hip -o build/var/${src}_$tag.o
objs=$(ls build/*.o | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/variants/libdpmn_$tag.so $objs build/var/${src}_$tag.o
echo built tools/variants/libdpmn_$tag.so
