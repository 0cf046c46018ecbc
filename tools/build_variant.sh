#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : libarchive_amd/csrc/libla_gpu_NAME.so = the shipped objects with la_lz4_inorder.hip
# (and la_lz4_fast.hip when FAST=1) recompiled with extra flags.  Diagnostic builds only (git-ignored); picked up through LA_GPU_LIB.
set -e
cd "$(dirname "$0")/../libarchive_amd/csrc"
name=$1; shift
make -s
HIPFLAGS="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -I../../include"
/opt/rocm/bin/hipcc $HIPFLAGS "$@" -c la_lz4_inorder.hip -o /tmp/la_lz4_inorder_$name.o
objs=$(ls *.o | grep -v la_lz4_inorder.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libla_gpu_$name.so /tmp/la_lz4_inorder_$name.o $objs
echo built libla_gpu_$name.so
