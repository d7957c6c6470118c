#!/bin/bash
# A second build of the library with extra compiler flags, for A/B timing on the GPU box:
#   tools/build_variant.sh NAME [flags...]   -> raytracer_amd/variants/librt_amd_NAME.so   (load with RT_AMD_LIB=<path>)
set -eu
NAME=${1:?name}; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p ../variants
make -s scene.o device_group.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xarch_device -fno-slp-vectorize "$@" -c pathtrace.hip -o ../variants/pathtrace_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/librt_amd_$NAME.so ../variants/pathtrace_$NAME.o scene.o device_group.o -lz -ldl -lpthread
rm -f ../variants/pathtrace_$NAME.o
echo built raytracer_amd/variants/librt_amd_$NAME.so
