#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : libarchive_amd/csrc/libla_gpu_NAME.so = the shipped objects with ONE source
# (SRC=..., default la_lz4_inorder.hip) recompiled with extra flags.  Diagnostic builds only (git-ignored); picked up through LA_GPU_LIB.
set -e
cd "$(dirname "$0")/../libarchive_amd/csrc"
name=$1; shift
src=${SRC:-la_lz4_inorder.hip}
base=${src%.hip}
make -s
HIPFLAGS="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -I../../include"
/opt/rocm/bin/hipcc $HIPFLAGS "$@" -c $src -o /tmp/${base}_$name.o
objs=$(ls *.o | grep -v "^$base.o\$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libla_gpu_$name.so /tmp/${base}_$name.o $objs
echo built libla_gpu_$name.so
