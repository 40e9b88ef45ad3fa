#!/bin/bash
# Build a variant of libemme_hip.so with extra -D flags on ONE kernel file (A/B on the GPU box through
# EMME_LIB=build/variants/<name>.so).  Usage: tools/build_variant.sh <name> <file.hip> "<flags>"
set -e
name=$1; file=$2; flags=$3
cd "$(dirname "$0")/../emme_amd/csrc"
make -s -j4 >/dev/null
mkdir -p ../../build/variants
obj=../../build/variants/${name}_$(basename $file .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $flags -c -o $obj $file
objs=$(ls build/*.o | grep -v "build/$(basename $file .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/$name.so $objs $obj -ldl
echo built build/variants/$name.so
