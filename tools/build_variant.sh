#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: libvb2.so with the kernels compiled under extra flags -> build_variants/NAME/libvb2.so
# (same-box A/B through VB2_LIB_PATH; build_variants/ is git-ignored but travels with gpurun snapshots)
set -e
cd "$(dirname "$0")/../verifybamid_amd/csrc"
name=$1; shift
out=../../build_variants/$name
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -mllvm -amdgpu-sched-strategy=iterative-ilp "$@" -c llk_kernels.hip -o $out/llk_kernels.o
objs=$(ls *.o | grep -v 'llk_kernels\|llk_passes')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libvb2.so $out/llk_kernels.o $objs -lpthread -lz -ldl
echo built $out/libvb2.so
