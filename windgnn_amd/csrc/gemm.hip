// fp32-input MFMA GEMM (v_mfma_f32_32x32x2_f32) used for the GRU projections and their gradients:
//   GI   = g   W_ih^T + b_ih     (forward input projection, torch nn.GRU called at
//                                 src/step6_gcn_gru_combined_model.py:23)
//   dg   = dGI W_ih              (backward through it)
//   dW_ih = dGI^T g, db_ih = dGI^T 1,  dW_hh = dGH^T Hprev, db_hh = dGH^T 1   (split-K over B*T)
// 128x128x16 tiles, 4 waves each owning 64x64 (2x2 MFMA tiles), operands staged k-major in LDS
// so that fragment reads are conflict-free.  Either operand may be K-contiguous or M/N-contiguous
// in global memory.  Exact fp32 (bitwise an fmaf chain along k).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 4;

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
template <bool IS_B>
__device__ __forceinline__ void fetch(float (&v)[8], const GemmP& p, const float* P, int ld, int kcontig, int r0,
                                      int nrows, int k0, int kend) {
  const int tid = threadIdx.x;
  if (kcontig) {
    const int r = tid >> 1, kc = (tid & 1) * 8;
    const int gr = r0 + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + kc + j;
      v[j] = (gr < nrows && k < kend) ? P[(size_t)gr * ld + k] : 0.f;
    }
  } else {
    const int kk = tid >> 4, rc = (tid & 15) * 8;
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
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int gr = r0 + rc + j;
      float x = 0.f;
      if (IS_B && p.ones_col && gr == nrows - 1) x = kvalid ? 1.f : 0.f;
      else if (krow && gr < ndata) x = P[rowoff + gr];
      v[j] = x;
    }
  }
}
__device__ __forceinline__ void commit(float* T, const float (&v)[8], int kcontig) {
  const int tid = threadIdx.x;
  if (kcontig) {
    const int r = tid >> 1, kc = (tid & 1) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) T[(kc + j) * LDT + r] = v[j];
  } else {
    const int kk = tid >> 4, rc = (tid & 15) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) T[kk * LDT + rc + j] = v[j];
  }
}

__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) float As[BK * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int z = blockIdx.z;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two-level summation: the MFMA chain over k is sequential, so every 512 k the running tile is folded into
  // `tot` (error grows with sqrt(512) + sqrt(K/512) instead of sqrt(K); K is 53 248 in the 4096-station
  // input projection).  For K <= 512 this is bitwise the single chain.
  f32x16 tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  int since = 0;
  float va[8], vb[8];
  if (kbeg < kend) {
    fetch<false>(va, p, p.A, p.lda, p.a_kc, m0, p.M, kbeg, kend);
    fetch<true>(vb, p, p.B, p.ldb, p.b_kc, n0, p.N, kbeg, kend);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    if (since == 512 / BK) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          tot[i][j] += acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
      since = 0;
    }
    ++since;
    commit(As, va, p.a_kc);
    commit(Bs, vb, p.b_kc);
    __syncthreads();
    if (k0 + BK < kend) {               // next step's operands: in flight under this step's MFMAs
      fetch<false>(va, p, p.A, p.lda, p.a_kc, m0, p.M, k0 + BK, kend);
      fetch<true>(vb, p, p.B, p.ldb, p.b_kc, n0, p.N, k0 + BK, kend);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a0 = As[(kk + lk) * LDT + wm + li];
      float a1 = As[(kk + lk) * LDT + wm + 32 + li];
      float b0 = Bs[(kk + lk) * LDT + wn + li];
      float b1 = Bs[(kk + lk) * LDT + wn + 32 + li];
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }

  float* C = p.partial ? p.partial + (size_t)z * p.M * p.N : p.C;
  const int ldc = p.partial ? p.N : p.ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
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

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int splitk, int M, int N,
                                     float* __restrict__ C, int ldc, int ncols_main, float* __restrict__ bias_out,
                                     const float* __restrict__ scales) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const size_t MN = (size_t)M * N;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  int z = 0;
  for (; z + 8 <= splitk; z += 8) {        // 8 independent loads in flight; fixed order => deterministic
    s0 += partial[(size_t)z * MN + i];
    s1 += partial[(size_t)(z + 1) * MN + i];
    s2 += partial[(size_t)(z + 2) * MN + i];
    s3 += partial[(size_t)(z + 3) * MN + i];
    s4 += partial[(size_t)(z + 4) * MN + i];
    s5 += partial[(size_t)(z + 5) * MN + i];
    s6 += partial[(size_t)(z + 6) * MN + i];
    s7 += partial[(size_t)(z + 7) * MN + i];
  }
  for (; z < splitk; ++z) s0 += partial[(size_t)z * MN + i];
  float s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  if (scales) s *= scales[1];
  int m = (int)(i / N), n = (int)(i % N);
  if (n < ncols_main) C[(size_t)m * ldc + n] = s;
  else if (bias_out && n == N - 1) bias_out[m] = s;
}

}  // namespace

int launch_gemm_f32(const GemmArgs& g, hipStream_t st) {
  GemmP p;
  p.A = g.A; p.lda = g.lda; p.a_kc = g.a_kcontig;
  p.B = g.B; p.ldb = g.ldb; p.b_kc = g.b_kcontig;
  p.C = g.C; p.ldc = g.ldc; p.M = g.M; p.N = g.N; p.K = g.K;
  p.bias = g.bias; p.ones_col = g.ones_col; p.shift_T = g.shift_T;
  p.splitk = g.splitk < 1 ? 1 : g.splitk;
  p.kchunk = cdiv_i(cdiv_i(g.K, p.splitk), BK) * BK;
  p.partial = g.partial;
  dim3 grid(cdiv_i(g.M, BM), cdiv_i(g.N, BN), p.splitk);
  const double fl = 2.0 * g.M * (double)g.N * g.K;
  const double by = 4.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N * p.splitk);
  PROF_LAUNCH("gemm_f32_kernel", fl, by, st, hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, st, p));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_splitk_reduce(const float* partial, int splitk, int M, int N, float* C, int ldc, int ncols_main,
                         float* bias_out, const float* scales, hipStream_t st) {
  size_t n = (size_t)M * N;
  PROF_LAUNCH("splitk_reduce_kernel", (double)n * splitk, 4.0 * n * (splitk + 1), st,
              hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial,
                                 splitk, M, N, C, ldc, ncols_main, bias_out, scales));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
