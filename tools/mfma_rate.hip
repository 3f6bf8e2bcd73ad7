// micro-benchmark: the MFMA rates the chip SUSTAINS (context for DESIGN.md sections 5 and 7: MFMA-dense kernels do not run
// at the 2.4 GHz the peak figures assume).  Every SIMD of every CU runs W waves that issue nothing but independent MFMAs
// out of registers for `iters` iterations; the launch is sized to last a few hundred microseconds and is repeated so the
// power controller reaches its steady state.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float* sink, int iters) {
  const float s = 1e-3f * (threadIdx.x & 7);
  if (KIND == 0) {          // v_mfma_f32_32x32x2_f32: 4096 flop, 64 cycles
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, a3, 0, 0, 0);
    }
    a0 += a1 + a2 + a3;
    if (a0[0] == 12345.f) sink[threadIdx.x] = a0[1];
  } else if (KIND == 1) {   // v_mfma_f32_16x16x4_f32: 2048 flop, 32 cycles
    f32x4 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s, s, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s, s, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(s, s, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(s, s, a3, 0, 0, 0);
    }
    a0 += a1 + a2 + a3;
    if (a0[0] == 12345.f) sink[threadIdx.x] = a0[1];
  } else if (KIND == 3) {   // v_mfma_f32_32x32x16_f16: 32768 flop, 32 cycles at the dense peak
    f16x8 h;
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)s;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a3, 0, 0, 0);
    }
    a0 += a1 + a2 + a3;
    if (a0[0] == 12345.f) sink[threadIdx.x] = a0[1];
  } else {                  // v_mfma_f32_16x16x32_f16: 16384 flop, 16 cycles at the dense peak
    f16x8 h;
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)s;
    f32x4 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a3, 0, 0, 0);
    }
    a0 += a1 + a2 + a3;
    if (a0[0] == 12345.f) sink[threadIdx.x] = a0[1];
  }
}

template <int KIND>
static void run(float* d, const char* name, double flop, double cycles, int wgs_per_cu, int iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float ms = 0.f;
  for (int rep = 0; rep < 8; ++rep) {      // back to back: the later repetitions are at the steady-state clock
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep == 0 || rep == 7) {
      const double n = 256.0 * wgs_per_cu * 4 * 4.0 * iters;   // MFMAs issued
      const double per_simd = n / 1024.0;
      printf("%-26s %d waves/SIMD rep %d: %7.1f us  %7.1f TFLOP/s  => %.2f GHz if one MFMA = %.0f cycles\n", name,
             wgs_per_cu, rep, ms * 1e3, n * flop / ms / 1e9, per_simd * cycles / (ms * 1e-3) / 1e9, cycles);
    }
  }
}

int main() {
  float* d; (void)hipMalloc(&d, 1 << 20);
  for (int w = 1; w <= 4; ++w) {
    run<0>(d, "v_mfma_f32_32x32x2_f32", 4096, 64, w, 2000 / w);
    run<1>(d, "v_mfma_f32_16x16x4_f32", 2048, 32, w, 4000 / w);
    run<2>(d, "v_mfma_f32_16x16x32_f16", 16384, 16, w, 8000 / w);
    run<3>(d, "v_mfma_f32_32x32x16_f16", 32768, 32, w, 4000 / w);
  }
  return 0;
}
