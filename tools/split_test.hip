// standalone check of the v_fma_mix-based fp16 hi/lo split against the C expression
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  float t0, t1;
  asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_cvt_pk_f16_f32 %1, %2, %3"
      : "=&v"(hi), "=&v"(lo), "=&v"(t0), "=&v"(t1)
      : "v"(x0), "v"(x1));
}
__global__ void k(const float* in, unsigned* out, _Float16* ref) {
  unsigned hi, lo;
  const int i = threadIdx.x;
  split2(in[2 * i], in[2 * i + 1], hi, lo);
  out[2 * i] = hi;
  out[2 * i + 1] = lo;
  for (int j = 0; j < 2; ++j) {
    float x = in[2 * i + j];
    _Float16 h = (_Float16)x;
    ref[4 * i + 2 * j] = h;
    ref[4 * i + 2 * j + 1] = (_Float16)(x - (float)h);
  }
}
int main() {
  const int n = 64;
  float h_in[2 * n];
  for (int i = 0; i < 2 * n; ++i) h_in[i] = (i % 7 - 3) * 0.37f + 1e-3f * i + (i == 5 ? 1e-7f : 0.f);
  float* d_in; unsigned* d_out; _Float16* d_ref;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, 2 * n * 4); hipMalloc(&d_ref, 4 * n * 2);
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, d_in, d_out, d_ref);
  unsigned h_out[2 * n]; unsigned short h_ref[4 * n];
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  hipMemcpy(h_ref, d_ref, sizeof(h_ref), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    unsigned hi = h_out[2 * i], lo = h_out[2 * i + 1];
    unsigned short a[4] = {(unsigned short)(hi & 0xffff), (unsigned short)(lo & 0xffff), (unsigned short)(hi >> 16), (unsigned short)(lo >> 16)};
    for (int j = 0; j < 4; ++j) if (a[j] != h_ref[4 * i + j]) { if (bad < 8) printf("mismatch i=%d j=%d got %04x ref %04x x=%g\n", i, j, a[j], h_ref[4 * i + j], h_in[2 * i + j / 2]); ++bad; }
  }
  printf("bad=%d\n", bad);
  return bad != 0;
}
