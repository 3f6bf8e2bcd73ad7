// micro-benchmark: what HBM rates the chip sustains for plain streaming (context for DESIGN.md section 7), and the
// per-CU global store rate.  256 workgroups; each iteration a workgroup writes ROWB bytes x 16 rows
// (rows strided like the gate stash), from NW waves, dwordx4 per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NW>
__global__ void __launch_bounds__(64 * NW) k(float* out, int iters, int row_floats, size_t row_stride, size_t step_stride) {
  const int tid = threadIdx.x;
  float* base = out + (size_t)blockIdx.x * 16 * row_stride;
  f32x4 v = {1.f * tid, 2.f, 3.f, 4.f};
  const int cpr = row_floats / 4;
  for (int t = 0; t < iters; ++t) {
    for (int q = tid; q < 16 * cpr; q += 64 * NW) {
      const int m = q / cpr, ch = q % cpr;
      *(f32x4*)(base + m * row_stride + t * step_stride + 4 * ch) = v;
    }
    __syncthreads();
  }
}
template <int NW> void run(float* d, const char* name, int row_floats, int nblk = 256) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 24;
  const size_t step = row_floats, row_stride = (size_t)iters * row_floats;
  if ((size_t)nblk * 16 * row_stride * 4 > ((size_t)1 << 30)) { printf("skipped: footprint beyond the 1 GiB buffer\n"); return; }
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k<NW>, dim3(nblk), dim3(64 * NW), 0, 0, d, iters, row_floats, row_stride, step);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)nblk * iters * 16 * row_floats * 4;
  printf("%s waves=%d row=%dB workgroups=%d: %.1f us, %.2f TB/s, %.1f B/clk/CU @2.2GHz\n", name, NW, row_floats * 4, nblk, ms * 1e3, bytes / ms / 1e9,
         bytes / nblk / (ms * 1e-3 * 2.2e9));
}
// streaming read (sum to keep the loads) and copy, 16 waves per CU-sized block, float4 per lane
__global__ void __launch_bounds__(1024) rd(const f32x4* in, size_t n4, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[0] = acc[0];
}
__global__ void __launch_bounds__(1024) cp(const f32x4* in, f32x4* out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
static void stream(float* d) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const size_t n4 = (size_t)256 * 1024 * 1024 / 16;   // 256 MB
  float ms;
  for (int rep = 0; rep < 3; ++rep) { hipEventRecord(a); hipLaunchKernelGGL(rd, dim3(2048), dim3(1024), 0, 0, (const f32x4*)d, n4, d + 4 * n4 + 64); hipEventRecord(b); hipEventSynchronize(b); }
  hipEventElapsedTime(&ms, a, b);
  printf("read  256 MB: %.1f us, %.2f TB/s\n", ms * 1e3, 256.0 * 1.048576 / ms / 1e3);
  for (int rep = 0; rep < 3; ++rep) { hipEventRecord(a); hipLaunchKernelGGL(cp, dim3(2048), dim3(1024), 0, 0, (const f32x4*)d, (f32x4*)d + n4 + 1024, n4); hipEventRecord(b); hipEventSynchronize(b); }
  hipEventElapsedTime(&ms, a, b);
  printf("copy  256 MB -> 256 MB: %.1f us, %.2f TB/s (read + write)\n", ms * 1e3, 512.0 * 1.048576 / ms / 1e3);
}
int main() {
  float* d; hipMalloc(&d, (size_t)1 << 30);
  stream(d);
  run<4>(d, "x4", 408); run<8>(d, "x4", 408); run<12>(d, "x4", 408); run<16>(d, "x4", 408);
  run<4>(d, "x4", 640); run<12>(d, "x4", 640);
  run<4>(d, "x4", 1024); run<12>(d, "x4", 1024);
  // fewer active CUs (one workgroup each): is ~11 B/clk/CU the CU's own store path or the chip's rate divided by 256?
  for (int nb : {16, 32, 64, 128, 256}) run<12>(d, "x4", 1024, nb);
  return 0;
}
