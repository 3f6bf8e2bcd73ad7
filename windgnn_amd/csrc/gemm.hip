// fp32-input MFMA GEMM (v_mfma_f32_32x32x2_f32) used for the GRU projections and their gradients:
//   GI   = g   W_ih^T + b_ih     (forward input projection, torch nn.GRU called at
//                                 src/step6_gcn_gru_combined_model.py:23)
//   dg   = dGI W_ih              (backward through it)
//   dW_ih = dGI^T g, db_ih = dGI^T 1,  dW_hh = dGH^T Hprev, db_hh = dGH^T 1   (split-K over B*T)
// 128x128x16 tiles, 4 waves each owning 64x64 (2x2 MFMA tiles), operands staged k-major in LDS
// so that fragment reads are conflict-free.  Either operand may be K-contiguous or M/N-contiguous
// in global memory.  Exact fp32 (bitwise an fmaf chain along k).
#include "common.h"
#include <cstdio>

namespace {

constexpr int BK = 16, LDT = 128 + 4;

struct GemmP {
  const float* A; int lda; int a_kc;
  const float* B; int ldb; int b_kc;
  float* C; int ldc;
  int M, N, K;
  const float* bias;
  int ones_col, shift_T, splitk, kchunk;
  float* partial;
};

// One thread's share (8 values) of a [BK][128] k-major LDS tile, fetched from a global operand into registers
// (fetch) and written to LDS later (commit), so that the loads of step k+1 fly under the MFMAs of step k.
// rows = m (or n) index, origin r0.  kcontig: element (r,k) at P[r*ld + k]; else at P[k*ld + r].
// NV consecutive floats at src, all in bounds: 16-byte loads when src allows it, 8-byte loads next, single dwords last
// (an odd row length, e.g. W_ih's 442, leaves every second row 8-byte aligned only)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV>
__device__ __forceinline__ void load_run(float (&v)[NV], const float* src) {
  const uintptr_t a = (uintptr_t)src;
  if ((a & 15) == 0) {
#pragma unroll
    for (int j = 0; j < NV; j += 4) {
      const f32x4 x = *(const f32x4*)(src + j);
      v[j] = x[0]; v[j + 1] = x[1]; v[j + 2] = x[2]; v[j + 3] = x[3];
    }
  } else if ((a & 7) == 0) {
#pragma unroll
    for (int j = 0; j < NV; j += 2) {
      const f32x2 x = *(const f32x2*)(src + j);
      v[j] = x[0]; v[j + 1] = x[1];
    }
  } else {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = src[j];
  }
}

// NV values per thread: the tile has 16 NV rows (128 for NV = 8; 64 for the short-M A tile, NV = 4).
template <bool IS_B, int NV>
__device__ __forceinline__ void fetch(float (&v)[NV], const GemmP& p, const float* P, int ld, int kcontig, int r0,
                                      int nrows, int k0, int kend) {
  const int tid = threadIdx.x;
  if (kcontig) {
    constexpr int TPR = BK / NV;            // threads per row
    const int r = tid / TPR, kc = (tid % TPR) * NV;
    const int gr = r0 + r;
    if (gr < nrows && k0 + kc + NV <= kend) {      // the whole run is in bounds: wide loads
      load_run<NV>(v, P + (size_t)gr * ld + k0 + kc);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int k = k0 + kc + j;
        v[j] = (gr < nrows && k < kend) ? P[(size_t)gr * ld + k] : 0.f;
      }
    }
  } else {
    const int kk = tid >> 4, rc = (tid & 15) * NV;
    const int k = k0 + kk;
    const bool kvalid = k < kend;
    bool krow = kvalid;                   // row carries data (the ones-column only needs kvalid)
    size_t rowoff = 0;
    if (IS_B && p.shift_T > 0) {          // Hprev: row k is Y row k-1, zero at window starts
      krow = kvalid && (k % p.shift_T) != 0;
      rowoff = krow ? (size_t)(k - 1) * ld : 0;
    } else {
      rowoff = krow ? (size_t)k * ld : 0;
    }
    const int ndata = nrows - (IS_B ? p.ones_col : 0);
    if (krow && r0 + rc + NV <= ndata) {           // the whole run is data of a live row: wide loads
      load_run<NV>(v, P + rowoff + r0 + rc);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int gr = r0 + rc + j;
        float x = 0.f;
        if (IS_B && p.ones_col && gr == nrows - 1) x = kvalid ? 1.f : 0.f;
        else if (krow && gr < ndata) x = P[rowoff + gr];
        v[j] = x;
      }
    }
  }
}
template <int NV>
__device__ __forceinline__ void commit(float* T, const float (&v)[NV], int kcontig) {
  const int tid = threadIdx.x;
  if (kcontig) {
    constexpr int TPR = BK / NV;
    const int r = tid / TPR, kc = (tid % TPR) * NV;
#pragma unroll
    for (int j = 0; j < NV; ++j) T[(kc + j) * LDT + r] = v[j];
  } else {
    const int kk = tid >> 4, rc = (tid & 15) * NV;
#pragma unroll
    for (int j = 0; j < NV; ++j) T[kk * LDT + rc + j] = v[j];
  }
}

// MI x NJ: 32 x 32 MFMA tiles per wave along M and N; the workgroup tile is 64 MI x 64 NJ.  2 x 2 = the 128 x 128
// tile; the 64-wide forms serve products whose 128-wide tiling would pad a dimension by more than the 64-wide one
// (N = 306 -> 320 instead of 384, 442 -> 448 instead of 512) or leave CUs without a workgroup (small batches:
// M = B*T = 6144 at BASELINE configs[1])
template <int MI, int NJ>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmP p) {
  constexpr int BMv = 64 * MI, BNv = 64 * NJ, NVA = 4 * MI, NVB = 4 * NJ;
  // two LDS stages: step k is multiplied out of stage k & 1 while step k+1's operands (loaded into registers during
  // step k-1) are committed to the other stage and step k+2's loads are issued -- ONE barrier per k step
  __shared__ __attribute__((aligned(16))) float As2[2][BK * LDT];
  __shared__ __attribute__((aligned(16))) float Bs2[2][BK * LDT];
  const int m0 = blockIdx.x * BMv, n0 = blockIdx.y * BNv;
  const int z = blockIdx.z;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 32 * MI, wn = (wave & 1) * 32 * NJ;
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two-level summation: the MFMA chain over k is sequential, so every 512 k the running tile is folded into
  // `tot` (error grows with sqrt(512) + sqrt(K/512) instead of sqrt(K); K is 53 248 in the 4096-station
  // input projection).  For K <= 512 this is bitwise the single chain.
  f32x16 tot[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  int since = 0;
  // RS register stages: while step i is multiplied out of LDS stage i & 1, the operands of steps i+1 .. i+RS are in
  // registers or in flight.  The 64-row tile (small M: few workgroups per CU to hide latency) takes two, the
  // 128-row tile one (it is at 244 VGPRs).
  constexpr int RS = MI * NJ == 4 ? 1 : 2;
  float va[RS][NVA], vb[RS][NVB];
  if (kbeg < kend) {
    fetch<false, NVA>(va[0], p, p.A, p.lda, p.a_kc, m0, p.M, kbeg, kend);
    fetch<true, NVB>(vb[0], p, p.B, p.ldb, p.b_kc, n0, p.N, kbeg, kend);
    commit<NVA>(As2[0], va[0], p.a_kc);
    commit<NVB>(Bs2[0], vb[0], p.b_kc);
#pragma unroll
    for (int u = 0; u < RS; ++u)
      if (kbeg + (1 + u) * BK < kend) {
        fetch<false, NVA>(va[u], p, p.A, p.lda, p.a_kc, m0, p.M, kbeg + (1 + u) * BK, kend);
        fetch<true, NVB>(vb[u], p, p.B, p.ldb, p.b_kc, n0, p.N, kbeg + (1 + u) * BK, kend);
      }
  }
  __syncthreads();
  int stg = 0;
  for (int kq = kbeg; kq < kend; kq += RS * BK) {
#pragma unroll
    for (int u = 0; u < RS; ++u) {
      const int k0 = kq + u * BK;
      if (k0 >= kend) break;            // workgroup-uniform
      if (since == 512 / BK) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            tot[i][j] += acc[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          }
        since = 0;
      }
      ++since;
      if (k0 + BK < kend) {             // step k+1: register stage u -> the other LDS stage
        commit<NVA>(As2[stg ^ 1], va[u], p.a_kc);
        commit<NVB>(Bs2[stg ^ 1], vb[u], p.b_kc);
      }
      if (k0 + (1 + RS) * BK < kend) {  // step k+1+RS: refill register stage u
        fetch<false, NVA>(va[u], p, p.A, p.lda, p.a_kc, m0, p.M, k0 + (1 + RS) * BK, kend);
        fetch<true, NVB>(vb[u], p, p.B, p.ldb, p.b_kc, n0, p.N, k0 + (1 + RS) * BK, kend);
      }
      const float* As = As2[stg];
      const float* Bs = Bs2[stg];
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float b[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = Bs[(kk + lk) * LDT + wn + 32 * j + li];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const float a = As[(kk + lk) * LDT + wm + 32 * i + li];
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(a, b[j], acc[i][j]);
        }
      }
      __syncthreads();                  // stage stg is free for step k+2; stage stg^1 is complete
      stg ^= 1;
    }
  }

  float* C = p.partial ? p.partial + (size_t)z * p.M * p.N : p.C;
  const int ldc = p.partial ? p.N : p.ldc;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wn + 32 * j + li;
      if (col >= p.N) continue;
      const float bv = (p.bias && !p.partial) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < p.M) C[(size_t)row * ldc + col] = (tot[i][j][r] + acc[i][j][r]) + bv;
      }
    }
}

}  // namespace

// Workgroup tile for an M x N product.  Measured (B = 4096 / 256, S = 34, H = 102): with thousands of tiles the 128 x 128
// form wins although it pads N = 306 to 384 (368 vs 408 us at 128 x 64: fewer operand bytes per MFMA); with few tiles the
// narrow forms win (M = 6144: 43 vs 59 us) -- they pad less and give every CU several workgroups.
void gemm_f32_tile(int M, int N, int* bm, int* bn) {
  if (cdiv_i(M, 128) * cdiv_i(N, 128) >= 1024) {
    *bm = *bn = 128;
    return;
  }
  *bn = cdiv_i(N, 64) * 64 < cdiv_i(N, 128) * 128 ? 64 : 128;
  *bm = (cdiv_i(M, 64) * 64 < cdiv_i(M, 128) * 128 || cdiv_i(M, 128) * cdiv_i(N, *bn) < 512) ? 64 : 128;
}
int gemm_f32_tiles(int M, int N) {
  int bm, bn;
  gemm_f32_tile(M, N, &bm, &bn);
  return cdiv_i(M, bm) * cdiv_i(N, bn);
}

// C[M][ldc] = sum_z partial[z][M][N] (+ bias[n]): the second half of a split-K NT product with few rows (the reference's own
// call shape, ONE window of 168 steps: 15 workgroups would otherwise each walk all of K; profiles/r5_dropin_latency.txt)
__global__ void nt_splitk_sum_kernel(const float* __restrict__ partial, int splitk, int M, int N, const float* __restrict__ bias,
                                     float* __restrict__ C, int ldc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  float s = partial[i];
  for (int z = 1; z < splitk; ++z) s += partial[(size_t)z * M * N + i];      // fixed order
  const int n = i % N;
  C[(size_t)(i / N) * ldc + n] = bias ? s + bias[n] : s;
}

// Split-K factor for an NT product C[M][N] with a contraction of K: only when the tiles leave most of the chip idle, chunks of
// at least 64 k, at most 8 of them.  1 = no split.
int gemm_f32_nt_splitk(int M, int N, int K) {
  if (gemm_f32_tiles(M, N) >= 64) return 1;
  int sk = K / 64;
  return sk < 2 ? 1 : (sk > 8 ? 8 : sk);
}

// NT product with the split decided by gemm_f32_nt_splitk: `kpart` >= splitk * M * N floats of scratch (unused when 1).
int launch_gemm_f32_nt(GemmArgs g, float* kpart, hipStream_t st) {
  const int sk = kpart ? gemm_f32_nt_splitk(g.M, g.N, g.K) : 1;
  if (sk == 1) { g.splitk = 1; g.partial = nullptr; return launch_gemm_f32(g, st); }
  const float* bias = g.bias;
  float* C = g.C;
  const int ldc = g.ldc;
  g.splitk = sk; g.partial = kpart; g.bias = nullptr;
  const int rc = launch_gemm_f32(g, st);
  if (rc != WGNN_OK) return rc;
  const int n = g.M * g.N;
  PROF_LAUNCH("nt_splitk_sum_kernel", 0.0, 4.0 * n * (sk + 1), st,
              hipLaunchKernelGGL(nt_splitk_sum_kernel, dim3(cdiv_i(n, 256)), dim3(256), 0, st, kpart, sk, g.M, g.N, bias, C, ldc));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_gemm_f32(const GemmArgs& g, hipStream_t st) {
  GemmP p;
  p.A = g.A; p.lda = g.lda; p.a_kc = g.a_kcontig;
  p.B = g.B; p.ldb = g.ldb; p.b_kc = g.b_kcontig;
  p.C = g.C; p.ldc = g.ldc; p.M = g.M; p.N = g.N; p.K = g.K;
  p.bias = g.bias; p.ones_col = g.ones_col; p.shift_T = g.shift_T;
  p.splitk = g.splitk < 1 ? 1 : g.splitk;
  p.kchunk = cdiv_i(cdiv_i(g.K, p.splitk), BK) * BK;
  p.partial = g.partial;
  const double fl = 2.0 * g.M * (double)g.N * g.K;
  const double by = 4.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N * p.splitk);
  int bm, bn;
  gemm_f32_tile(g.M, g.N, &bm, &bn);
  if (bm == 128 && cdiv_i(g.M, 128) * cdiv_i(g.N, bn) * p.splitk < 256) bm = 64;   // a split-K chosen elsewhere
  dim3 grid(cdiv_i(g.M, bm), cdiv_i(g.N, bn), p.splitk);
  // profile name: tile and operand forms ([kk] = both K-contiguous, [kn], [tn] = A given transposed)
  char name[64];
  snprintf(name, sizeof name, "gemm_f32_kernel<%d,%d>[%s%s%s]", bm, bn, g.a_kcontig ? (g.b_kcontig ? "kk" : "kn") : "tn",
           g.ones_col ? ",ones" : "", g.shift_T > 0 ? ",shift" : "");
  if (bm == 128 && bn == 128) {
    PROF_LAUNCH(name, fl, by, st, hipLaunchKernelGGL((gemm_f32_kernel<2, 2>), grid, dim3(256), 0, st, p));
  } else if (bm == 128) {
    PROF_LAUNCH(name, fl, by, st, hipLaunchKernelGGL((gemm_f32_kernel<2, 1>), grid, dim3(256), 0, st, p));
  } else if (bn == 128) {
    PROF_LAUNCH(name, fl, by, st, hipLaunchKernelGGL((gemm_f32_kernel<1, 2>), grid, dim3(256), 0, st, p));
  } else {
    PROF_LAUNCH(name, fl, by, st, hipLaunchKernelGGL((gemm_f32_kernel<1, 1>), grid, dim3(256), 0, st, p));
  }
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
