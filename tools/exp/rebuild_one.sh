#!/bin/bash
# rebuild ONE source of the library with extra flags and relink (A/B experiments on the GPU box): rebuild_one.sh grux.hip -DX=1
set -e
cd "$(dirname "$0")/../../windgnn_amd/csrc"
src=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -I../../include "$@" -c $src -o ${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o libwindgnn_hip.so
