#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-DFLAG ...]: vhap_amd/lib/libvhap_hip_NAME.so = the current library with FILE.hip recompiled with the extra flags
# (A/B of kernel variants on one GPU box: VHAP_HIP_LIB, tools/ab_libs.sh)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m vhap_amd.build > /dev/null
obj=vhap_amd/lib/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Iinclude -Ivhap_amd/csrc "$@" -x hip -c vhap_amd/csrc/$src -o $obj
objs=$(ls vhap_amd/lib/*.o | grep -v "_[a-zA-Z0-9]*\.o$" | grep -v "/${src%.hip}\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vhap_amd/lib/libvhap_hip_$name.so $objs $obj
echo vhap_amd/lib/libvhap_hip_$name.so
