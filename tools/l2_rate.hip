// micro-benchmark: how fast ONE CU can pull L2-resident data, by load path (context for DESIGN.md section 5: the plane
// GEMMs are bound by their L2 -> LDS staging).  256 workgroups of W waves; each reads `iters` blocks of 64 KB out of a
// 2 MB window (L2-resident after the first touch) with 16 bytes per lane:
//   regs     global_load_dwordx4 -> VGPRs (summed)
//   regs+lds global_load_dwordx4 -> VGPRs -> ds_write_b128
//   dma      global_load_lds_dwordx4 (LDS-DMA, what pgemm.hip uses)
//   dma64    LDS-DMA of 64-byte row segments at a 896-byte row stride (the NT GEMM's A operand: 32 halfs of K per row
//            and stage), over an L2-resident window and over 1 GB (HBM)
//   hipcc --offload-arch=gfx950 -O3 tools/l2_rate.hip -o /tmp/l2_rate && /tmp/l2_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

template <int MODE, int W>
__global__ void __launch_bounds__(64 * W) k(const float* __restrict__ buf, size_t win_floats, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int BLK = 16384;                    // floats per 64 KB block
  constexpr int PER = BLK / (64 * W * 4);       // 16-byte loads per lane per block
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t nblk = win_floats / BLK;
  for (int it = 0; it < iters; ++it) {
    const float* src = buf + ((size_t)(blockIdx.x * 7 + it * 13) % nblk) * BLK;
    if (MODE == 3) {       // 16 rows x 64 B per wave-instruction; a block = 1024 rows x 64 B of a [rows][896 B] matrix
      const size_t rows_total = win_floats * 4 / 896;
      const size_t r0 = ((size_t)(blockIdx.x * 7 + it * 13) * 1024) % (rows_total - 1024);
      const int kseg = it % 14;                  // which 64-byte segment of the rows
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        const size_t row = r0 + (size_t)(p * W + wave) * 16 + (lane >> 2);
        __builtin_amdgcn_global_load_lds((glb_void*)((const char*)buf + row * 896 + kseg * 64 + (lane & 3) * 16),
                                         (lds_void*)(smem + (p * W + wave) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 2) {
#pragma unroll
      for (int p = 0; p < PER; ++p)
        __builtin_amdgcn_global_load_lds((glb_void*)(src + (size_t)(p * W + wave) * 256 + lane * 4),
                                         (lds_void*)(smem + (p * W + wave) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      f32x4 v[PER];
#pragma unroll
      for (int p = 0; p < PER; ++p) v[p] = *(const f32x4*)(src + (size_t)(p * W + wave) * 256 + lane * 4);
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        if (MODE == 1) *(f32x4*)(smem + (p * W + wave) * 1024 + lane * 16) = v[p];
        else acc += v[p];
      }
    }
  }
  if (MODE != 0) {
    __syncthreads();
    acc = *(const f32x4*)(smem + tid * 16);
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[tid] = acc[0];
}

template <int MODE, int W>
static void run(const float* d, float* sink, const char* name, size_t win_bytes = (size_t)2 << 20) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 400;
  const size_t win = win_bytes >> 2;            // window in floats
  float ms = 0.f;
  (void)hipFuncSetAttribute((const void*)k<MODE, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, W>), dim3(256), dim3(64 * W), 65536, 0, d, win, iters, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
  }
  const double bytes = 256.0 * iters * 65536.0;
  printf("%-9s %2d waves/CU: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU @2.1GHz\n", name, W, ms * 1e3, bytes / ms / 1e9,
         bytes / 256 / (ms * 1e-3 * 2.1e9));
}

int main() {
  float *d, *sink;
  (void)hipMalloc(&d, (size_t)1 << 30);
  (void)hipMalloc(&sink, 1 << 16);
  (void)hipMemset(d, 0, (size_t)1 << 30);
  run<0, 4>(d, sink, "regs"); run<0, 8>(d, sink, "regs"); run<0, 16>(d, sink, "regs");
    run<2, 4>(d, sink, "dma"); run<2, 8>(d, sink, "dma"); run<2, 16>(d, sink, "dma");
  run<2, 8>(d, sink, "dma 1GB", (size_t)1 << 30); run<2, 16>(d, sink, "dma 1GB", (size_t)1 << 30);
  run<3, 8>(d, sink, "dma64"); run<3, 16>(d, sink, "dma64");
  run<3, 8>(d, sink, "dma64 1GB", (size_t)1 << 30); run<3, 16>(d, sink, "dma64 1GB", (size_t)1 << 30);
  return 0;
}
