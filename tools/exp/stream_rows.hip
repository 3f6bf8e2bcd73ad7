// Scratch micro-benchmark (not product code): what the read / write path gives for the GCN kernels' access shape --
// one wave per [S*13]-float tile row, persistent grid, 16 waves per CU -- so that the kernels' "memory skeleton" time can be
// split into its parts.  Build: hipcc --offload-arch=gfx950 -O3 tools/exp/stream_rows.hip -o gpurun_out/stream_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE bit 0: 16-byte loads (else 8-byte); bit 1: stage through LDS (ds_write_b32 scatter, rows of 20 words) and read back;
// bit 2: store two fp16-sized planes (dword per lane) ; bit 3: stores 16 B per lane
template <int DEPTH, int MODE>
__global__ __launch_bounds__(512, 2) void rows_kernel(int ntiles, int I, int ldp, const float* __restrict__ X,
                                                       unsigned* __restrict__ P0, unsigned* __restrict__ P1, float* sink) {
  __shared__ float lds[8 * 48 * 20];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave_id = blockIdx.x * 8 + w, nwaves = gridDim.x * 8;
  float* xb = lds + w * 48 * 20;
  constexpr bool WIDE = MODE & 1, STAGE = MODE & 2, STORE = MODE & 4, WSTORE = MODE & 8, NTS = MODE & 16, NTL = MODE & 32;
  constexpr int NP = 5, NQ = 3;
  int o[NP][2];
  for (int k = 0; k < NP; ++k) {
    int e = 2 * (lane + 64 * k);
    if (e + 1 >= I) e = I - 2;
    o[k][0] = (e / 13) * 20 + e % 13;
    o[k][1] = ((e + 1) / 13) * 20 + (e + 1) % 13;
  }
  const int npairs = I / 2, nquads = (I + 3) / 4;
  float acc = 0.f;
  f32x2 r2[DEPTH][NP];
  f32x4 r4[DEPTH][NQ];
  auto load = [&](int t, int slot) {
    const float* src = X + (size_t)t * I;
    if (WIDE) {
#pragma unroll
      for (int k = 0; k < NQ; ++k)
        if (64 * k < nquads) { int q = lane + 64 * k; int e = q < I / 4 ? 4 * q : I - 4; r4[slot][k] = NTL ? __builtin_nontemporal_load((const f32x4u*)(src + e)) : *(const f32x4u*)(src + e); }
    } else {
#pragma unroll
      for (int k = 0; k < NP; ++k)
        if (64 * k < npairs) { int p = lane + 64 * k; if (p >= npairs) p = npairs - 1; r2[slot][k] = NTL ? __builtin_nontemporal_load((const f32x2*)(src + 2 * p)) : *(const f32x2*)(src + 2 * p); }
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (wave_id + d * nwaves < ntiles) load(wave_id + d * nwaves, d);
  int it = 0;
  for (int tile = wave_id; tile < ntiles; tile += DEPTH * nwaves) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = tile + d * nwaves;
      if (t >= ntiles) break;
      float v[2 * NP];
      if (WIDE) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) if (64 * k < nquads) { v[0] += r4[d][k][0] + r4[d][k][2]; v[1] += r4[d][k][1] + r4[d][k][3]; }
#pragma unroll
        for (int k = 0; k < NP; ++k) { v[2 * k] = r4[d][k % NQ][0]; v[2 * k + 1] = r4[d][k % NQ][1]; }
      } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) { v[2 * k] = r2[d][k][0]; v[2 * k + 1] = r2[d][k][1]; }
      }
      if (t + DEPTH * nwaves < ntiles) load(t + DEPTH * nwaves, d);
      if (STAGE) {
#pragma unroll
        for (int k = 0; k < NP; ++k)
          if (64 * k < npairs) { xb[o[k][0]] = v[2 * k]; xb[o[k][1]] = v[2 * k + 1]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NP; ++k)
          if (64 * k < npairs) { v[2 * k] = xb[o[k][1]]; v[2 * k + 1] = xb[o[k][0]]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
      if (STORE) {
        if (WSTORE) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          if (lane < ldp / 8) {
            u32x4 a = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            u32x4 b = {__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
            if (NTS) { __builtin_nontemporal_store(a, (u32x4*)(P0 + (size_t)t * (ldp / 2) + 4 * lane)); __builtin_nontemporal_store(b, (u32x4*)(P1 + (size_t)t * (ldp / 2) + 4 * lane)); }
            else { *(u32x4*)(P0 + (size_t)t * (ldp / 2) + 4 * lane) = a; *(u32x4*)(P1 + (size_t)t * (ldp / 2) + 4 * lane) = b; }
          }
        } else {
#pragma unroll
          for (int k = 0; k < NP; ++k)
            if (64 * k < ldp / 2) {
              int p = lane + 64 * k;
              if (p < ldp / 2) { if (NTS) { __builtin_nontemporal_store(__float_as_uint(v[2 * k]), &P0[(size_t)t * (ldp / 2) + p]); __builtin_nontemporal_store(__float_as_uint(v[2 * k + 1]), &P1[(size_t)t * (ldp / 2) + p]); } else { P0[(size_t)t * (ldp / 2) + p] = __float_as_uint(v[2 * k]); P1[(size_t)t * (ldp / 2) + p] = __float_as_uint(v[2 * k + 1]); } }
            }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 2 * NP; ++k) acc += v[k];
      }
    }
    ++it;
  }
  if (acc == 12345.678f) sink[wave_id] = acc + it;
}

static int g_flush = 1;
template <int DEPTH, int MODE>
void run(const char* name, int ntiles, int I, int ldp, const float* X, unsigned* P0, unsigned* P1, float* sink) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const int grid = 512;
  static void* flush = nullptr;
  if (!flush) CHECK(hipMalloc(&flush, (size_t)768 << 20));
  for (int i = 0; i < 3; ++i) rows_kernel<DEPTH, MODE><<<grid, 512>>>(ntiles, I, ldp, X, P0, P1, sink);
  const int reps = 10;
  double us = 0;
  for (int i = 0; i < reps; ++i) {
    if (g_flush == 1) CHECK(hipMemsetAsync(flush, i, (size_t)768 << 20));      // dirty lines fill the Infinity Cache
    if (g_flush == 2) rows_kernel<2, 1><<<grid, 512>>>(768 * 256, 1024, 0, (const float*)flush, P0, P1, sink);  // clean lines
    CHECK(hipEventRecord(a));
    rows_kernel<DEPTH, MODE><<<grid, 512>>>(ntiles, I, ldp, X, P0, P1, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    us += ms * 1e3 / reps;
  }
  double bytes = (double)ntiles * I * 4 + ((MODE & 4) ? (double)ntiles * ldp * 4 : 0.0);
  printf("%-44s %7.1f us  %6.2f TB/s\n", name, us, bytes / us * 1e-6);
}

int main() {
  const int ntiles = 4096 * 24, I = 442, ldp = 448;
  float *X, *sink; unsigned *P0, *P1;
  CHECK(hipMalloc(&X, (size_t)ntiles * I * 4 + 64));
  CHECK(hipMalloc(&P0, (size_t)ntiles * ldp * 2));
  CHECK(hipMalloc(&P1, (size_t)ntiles * ldp * 2));
  CHECK(hipMalloc(&sink, 1 << 20));
  CHECK(hipMemset(X, 0, (size_t)ntiles * I * 4));
#define RUN(D, M, name) run<D, M>(name, ntiles, I, ldp, X, P0, P1, sink)
  for (g_flush = 0; g_flush < 3; ++g_flush) {
    printf("-- before each timed launch: %s\n", g_flush == 0 ? "nothing (X and P resident in the Infinity Cache)" : g_flush == 1 ? "768 MB memset (dirty lines)" : "768 MB read (clean lines)");
    RUN(1, 0, "read 8B/lane depth1");
    RUN(1, 32, "read 8B/lane depth1, nt loads");
    RUN(1, 4, "read 8B d1 + store 4B/lane");
    RUN(1, 4 + 16, "read 8B d1 + store 4B/lane, nt stores");
    RUN(1, 4 + 16 + 32, "read 8B d1 + store 4B/lane, nt both");
    RUN(2, 13, "read 16B d2 + store 16B");
    RUN(2, 13 + 16, "read 16B d2 + store 16B, nt stores");
    RUN(2, 13 + 16 + 32, "read 16B d2 + store 16B, nt both");
  }
  return 0;
}
