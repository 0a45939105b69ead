#!/bin/bash
# One more build of the Float32 library with extra compiler definitions, for A/B runs on one box:
#   tools/build_variant.sh NAME -DGB25_FOO=1 ...   ->  ab/libgb25hip_NAME.so   (ab/ travels with gpurun; git-ignored like every .so)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $REPO/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-pass-failed -fno-slp-vectorize \
  -mllvm -amdgpu-use-amdgpu-trackers -DGB25_REAL=float "$@" -o $REPO/ab/libgb25hip_$NAME.so $REPO/gb-25_amd/csrc/gb25_api.hip -ldl
echo built ab/libgb25hip_$NAME.so
