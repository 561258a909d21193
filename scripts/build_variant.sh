#!/bin/bash
# scripts/build_variant.sh <name> <file.hip> [-DFLAG ...]: one translation unit rebuilt with extra flags, linked with the
# tree's other objects into _variants/lib_<name>.so (select it with LSA_LIB=...)
name=$1; src=$2; shift 2
cd lidarslam_amd/csrc
mkdir -p _build/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -c $src -o _build/var/$name.o || exit 1
objs=$(ls _build/*.o _build/host/*.o | grep -v "_build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../_variants/lib_$name.so $objs _build/var/$name.o
echo built _variants/lib_$name.so
