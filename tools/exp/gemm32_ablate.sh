#!/bin/bash
# Exact-fp32 NT GEMMs (GI, dg at B = 4096): where the time beyond the MFMA floor goes.  Builds variants of the library that
# differ in gemm32.hip only (G32_ABLATE: 1 no C stores, 2 no DMA after the first stage, 4 no barrier / wait) HERE (cross-compile)
# into build_ab/, then `python tools/exp/ab_step.py --math f32 build_ab/*.so` on the GPU box times them in alternating processes.
set -e
cd "$(dirname "$0")/../.."
mkdir -p build_ab
for v in 0 1 2 3 6 7; do
  bash tools/exp/rebuild_one.sh gemm32.hip -DG32_ABLATE=$v
  cp windgnn_amd/csrc/libwindgnn_hip.so build_ab/g32_ablate_$v.so
done
bash tools/exp/rebuild_one.sh gemm32.hip
