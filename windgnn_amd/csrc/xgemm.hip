// Split-fp16 ("f16x3") MFMA GEMMs for the GRU projections and their gradients.
//
// Every fp32 operand x is split on the fly into two fp16 values hi = fp16(x), lo = fp16(x - hi)
// (22 significant bits together) and each product is formed as hi*hi + hi*lo + lo*hi with
// v_mfma_f32_32x32x16_f16 accumulating in fp32: three MFMA passes at the fp16 rate (16x the fp32
// MFMA rate) with fp32-grade error (measured ~6e-6 on Y, DESIGN.md "Numerics").  Operands whose
// magnitude is far from 1 (the backward's dGI ~ 1e-7) are pre-multiplied by a power of two taken
// from max|dY| and the result is un-scaled in the epilogue, so fp16's exponent range is never hit.
//
// Roles (reference: nn.GRU at src/step6_gcn_gru_combined_model.py:23 and its autograd, src/main.py:79)
//   NT  GI = g W_ih^T + b_ih            A fp32 [M][Kp] rows, B pre-split planes [BN][Kp]
//   NT  dg = dGI W_ih                   same kernel, B = split(W_ih^T)
//   TN  dW_ih = dGI^T [g | 1]           both operands fp32 [K][cols]; tiles are gathered k-major
//   TN  dW_hh = dGH^T [Hprev | 1]       B rows shifted by one timestep, zero at window starts
//
// LDS tiles are [row][32 k] fp16 (64-B rows) with the 16-B chunk index XOR-swizzled by
// (row>>2)&3, which makes the ds_read_b128 fragment reads of v_mfma_f32_32x32x16_f16 conflict-free.
#include <stdlib.h>

#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vectors: HIP's uint4/float4 structs defeat SROA

namespace {

constexpr int XT = 512;   // 8 waves

__device__ __forceinline__ int sw_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ void split8(const float* x, float s, h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = x[j] * s;
    const _Float16 h = (_Float16)v;
    hi[j] = h;
    lo[j] = (_Float16)(v - (float)h);
  }
}

__device__ __forceinline__ f32x16 mfma_h(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// one 32-deep K stage for a wave tile of MT_W x NT_W 32x32 tiles
template <int MT_W, int NT_W>
__device__ __forceinline__ void compute_stage(const char* Ahi, const char* Alo, const char* Bhi, const char* Blo,
                                              int wm0, int wn0, f32x16 (&acc)[MT_W][NT_W], int lane, int nt_valid) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = 2 * ks + lh;
    h8 ah[MT_W], al[MT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i) {
      const int off = sw_off(wm0 + 32 * i + li, c);
      ah[i] = *(const h8*)(Ahi + off);
      al[i] = *(const h8*)(Alo + off);
    }
#pragma unroll
    for (int j = 0; j < NT_W; ++j) {
      if (j < nt_valid) {
        const int off = sw_off(wn0 + 32 * j + li, c);
        const h8 bh = *(const h8*)(Bhi + off);
        const h8 bl = *(const h8*)(Blo + off);
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
          acc[i][j] = mfma_h(al[i], bh, acc[i][j]);
          acc[i][j] = mfma_h(ah[i], bl, acc[i][j]);
          acc[i][j] = mfma_h(ah[i], bh, acc[i][j]);
        }
      }
    }
  }
}

template <int NB>
__device__ __forceinline__ void nt_gload(bool arow_ok, const float* aptr, const _Float16* Bg, const int (&b_src)[NB],
                                         int kt_stride, int k0, f32x4& av0, f32x4& av1, u32x4 (&breg)[NB]) {
  if (arow_ok) {
    av0 = *(const f32x4*)(aptr + k0);
    av1 = *(const f32x4*)(aptr + k0 + 4);
  } else {
    av0 = (f32x4){0.f, 0.f, 0.f, 0.f};
    av1 = av0;
  }
#pragma unroll
  for (int it = 0; it < NB; ++it) breg[it] = *(const u32x4*)(Bg + b_src[it] + (size_t)(k0 >> 5) * kt_stride);
}

template <int NB>
__device__ __forceinline__ void nt_swrite(char* st, int a_dst, int lo_off, const int (&b_dst)[NB], float s_in,
                                          const f32x4& av0, const f32x4& av1, const u32x4 (&breg)[NB]) {
  const float xs[8] = {av0[0], av0[1], av0[2], av0[3], av1[0], av1[1], av1[2], av1[3]};
  h8 hi, lo;
  split8(xs, s_in, hi, lo);
  *(h8*)(st + a_dst) = hi;
  *(h8*)(st + lo_off + a_dst) = lo;
#pragma unroll
  for (int it = 0; it < NB; ++it) *(u32x4*)(st + b_dst[it]) = breg[it];
}

// ------------------------------------------------------------------------------------------------
// NT: C[M][N] = scale_out * (scale_in*A)[M][Kp] . B[BN][Kp]^T + bias.   BM = 128, BN = 64*NT_W.
template <int NT_W>
__global__ void __launch_bounds__(XT) xgemm_nt_kernel(const float* __restrict__ A, int lda, int M, int Kp,
                                                      const _Float16* __restrict__ Bhi_g,
                                                      const _Float16* __restrict__ Blo_g, float* __restrict__ C,
                                                      int ldc, int N, const float* __restrict__ bias,
                                                      const float* __restrict__ s_in_p,
                                                      const float* __restrict__ s_out_p, int dbg) {
  (void)Blo_g;   // the lo plane follows the hi plane: Blo_g == Bhi_g + BN*Kp
  constexpr int BM = 128, BN = 64 * NT_W, STAGE = (2 * BM + 2 * BN) * 64, NB = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave & 3) * 32, wn0 = (wave >> 2) * 32 * NT_W;
  const int m0 = blockIdx.x * BM;
  const float s_in = s_in_p ? s_in_p[0] : 1.f, s_out = s_out_p ? s_out_p[1] : 1.f;

  f32x16 acc[1][NT_W];
#pragma unroll
  for (int j = 0; j < NT_W; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  const int arow = tid >> 2, akc = tid & 3;
  const bool arow_ok = (m0 + arow) < M;
  const float* aptr = A + (size_t)(arow_ok ? m0 + arow : 0) * lda + 8 * akc;
  // B-plane chunk owned by this thread in iteration `it`: q = tid + XT*it
  int b_src[NB], b_dst[NB];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    const int q = tid + XT * it;
    const int plane = q / (BN * 4), rem = q - plane * (BN * 4);
    const int row = rem >> 2, c = rem & 3;
    b_src[it] = plane * BN * Kp + row * 32 + 8 * c;          // halfs; planes are stage-major [kt][BN][32], lo after hi
    b_dst[it] = 2 * BM * 64 + plane * BN * 64 + sw_off(row, c);
  }
  const int a_dst = sw_off(arow, akc);
  // Two register staging sets: the loads of stage kt+2 are issued while stage kt computes, so each
  // load has two full compute phases to land (the kernel was load-latency-bound with one).
  f32x4 a0A, a1A, a0B, a1B;
  u32x4 bA[NB], bB[NB];
  const int nk = Kp / 32;
  const int KTS = BN * 32;
  nt_gload<NB>(arow_ok, aptr, Bhi_g, b_src, KTS, 0, a0A, a1A, bA);
  nt_swrite<NB>(smem, a_dst, BM * 64, b_dst, s_in, a0A, a1A, bA);
  if (nk > 1) nt_gload<NB>(arow_ok, aptr, Bhi_g, b_src, KTS, 32, a0A, a1A, bA);      // stage 1 -> set A
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    {  // even stage kt: compute buf0; prefetch kt+2 into set B; then stage kt+1 (set A) -> buf1
      if (kt + 2 < nk && !(dbg & 1)) nt_gload<NB>(arow_ok, aptr, Bhi_g, b_src, KTS, 32 * (kt + 2), a0B, a1B, bB);
      if (!(dbg & 2))
        compute_stage<1, NT_W>(smem, smem + BM * 64, smem + 2 * BM * 64, smem + 2 * BM * 64 + BN * 64, wm0, wn0, acc,
                               lane, NT_W);
      if (kt + 1 < nk) nt_swrite<NB>(smem + STAGE, a_dst, BM * 64, b_dst, s_in, a0A, a1A, bA);
      __syncthreads();
    }
    if (kt + 1 < nk) {  // odd stage kt+1: compute buf1; prefetch kt+3 into set A; then stage kt+2 (set B) -> buf0
      const char* cur = smem + STAGE;
      if (kt + 3 < nk && !(dbg & 1)) nt_gload<NB>(arow_ok, aptr, Bhi_g, b_src, KTS, 32 * (kt + 3), a0A, a1A, bA);
      if (!(dbg & 2))
        compute_stage<1, NT_W>(cur, cur + BM * 64, cur + 2 * BM * 64, cur + 2 * BM * 64 + BN * 64, wm0, wn0, acc, lane,
                               NT_W);
      if (kt + 2 < nk) nt_swrite<NB>(smem, a_dst, BM * 64, b_dst, s_in, a0B, a1B, bB);
      __syncthreads();
    }
  }

  const int li = lane & 31, lh = lane >> 5;
  if (dbg & 16) return;
#pragma unroll
  for (int j = 0; j < NT_W; ++j) {
    const int col = wn0 + 32 * j + li;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < M) C[(size_t)row * ldc + col] = acc[0][j][r] * s_out + bv;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TN: partial[z][Mout][Nout] = sum_{k in chunk z} (scale_in*A[k][m]) * Bv[k][n]
//   Bv[k][n] = 1 if n == ones_col; else B[krow(k)][n] for n < ncols_b, where krow(k) = k, or with
//   shift_T > 0: k-1 and the whole row is 0 when k % shift_T == 0 (Hprev = Y shifted one step).
// BM = 64*MT_W covers all of M in one block; BN = 128*NT_W; grid = (splitk, ceil(N/BN)).
template <int MT_W, int NT_W>
__global__ void __launch_bounds__(XT) xgemm_tn_kernel(const float* __restrict__ A, int lda, int mcols,
                                                      const float* __restrict__ B, int ldb, int ncols_b,
                                                      int ones_col, int shift_T, int K, int kchunk,
                                                      float* __restrict__ partial, int Mout, int Nout,
                                                      const float* __restrict__ s_in_p) {
  constexpr int BM = 64 * MT_W, BN = 128 * NT_W, STAGE = (2 * BM + 2 * BN) * 64;
  constexpr int UA = (BM * 4 + XT - 1) / XT, UB = (BN * 4 + XT - 1) / XT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave & 1) * 32 * MT_W, wn0 = (wave >> 1) * 32 * NT_W;
  const int n0 = blockIdx.y * BN;   // blocks z + splitk*y: same z => ids differ by a multiple of splitk => one XCD
  const int z = blockIdx.x;
  const int kbeg = z * kchunk, kend = min(K, kbeg + kchunk);
  const float s_in = s_in_p ? s_in_p[0] : 1.f;

  f32x16 acc[MT_W][NT_W];
#pragma unroll
  for (int i = 0; i < MT_W; ++i)
#pragma unroll
    for (int j = 0; j < NT_W; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // how many of this wave's n-tiles hold any real output column
  int nt_valid = 0;
#pragma unroll
  for (int j = 0; j < NT_W; ++j)
    if (n0 + wn0 + 32 * j < Nout) nt_valid = j + 1;

  float areg[UA][8], breg[UB][8];
#define TN_GLOAD(k0)                                                                         \
  do {                                                                                       \
    _Pragma("unroll") for (int u = 0; u < UA; ++u) {                                         \
      const int unit = tid + XT * u;                                                         \
      const int m = unit % BM, kc = unit / BM;                                               \
      const bool ok = unit < BM * 4 && m < mcols;                                            \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                        \
        const int k = (k0) + 8 * kc + j;                                                     \
        areg[u][j] = (ok && k < kend) ? A[(size_t)k * lda + m] : 0.f;                        \
      }                                                                                      \
    }                                                                                        \
    _Pragma("unroll") for (int u = 0; u < UB; ++u) {                                         \
      const int unit = tid + XT * u;                                                         \
      const int nl = unit % BN, kc = unit / BN;                                              \
      const int n = n0 + nl;                                                                 \
      const bool inb = unit < BN * 4;                                                        \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                        \
        const int k = (k0) + 8 * kc + j;                                                     \
        float v = 0.f;                                                                       \
        if (inb && k < kend) {                                                               \
          if (n == ones_col) v = 1.f;                                                        \
          else if (n < ncols_b) {                                                            \
            if (shift_T > 0) { if (k % shift_T != 0) v = B[(size_t)(k - 1) * ldb + n]; }     \
            else v = B[(size_t)k * ldb + n];                                                 \
          }                                                                                  \
        }                                                                                    \
        breg[u][j] = v;                                                                      \
      }                                                                                      \
    }                                                                                        \
  } while (0)
#define TN_SWRITE(st)                                                                        \
  do {                                                                                       \
    _Pragma("unroll") for (int u = 0; u < UA; ++u) {                                         \
      const int unit = tid + XT * u;                                                         \
      if (unit < BM * 4) {                                                                   \
        const int m = unit % BM, kc = unit / BM;                                             \
        h8 hi, lo;                                                                           \
        split8(areg[u], s_in, hi, lo);                                                       \
        const int off = sw_off(m, kc);                                                       \
        *(h8*)((st) + off) = hi;                                                             \
        *(h8*)((st) + BM * 64 + off) = lo;                                                   \
      }                                                                                      \
    }                                                                                        \
    _Pragma("unroll") for (int u = 0; u < UB; ++u) {                                         \
      const int unit = tid + XT * u;                                                         \
      if (unit < BN * 4) {                                                                   \
        const int nl = unit % BN, kc = unit / BN;                                            \
        h8 hi, lo;                                                                           \
        split8(breg[u], 1.f, hi, lo);                                                        \
        const int off = sw_off(nl, kc);                                                      \
        *(h8*)((st) + 2 * BM * 64 + off) = hi;                                               \
        *(h8*)((st) + 2 * BM * 64 + BN * 64 + off) = lo;                                     \
      }                                                                                      \
    }                                                                                        \
  } while (0)

  const int nk = (kend - kbeg + 31) / 32;
  if (nk > 0) {
    TN_GLOAD(kbeg);
    TN_SWRITE(smem);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * STAGE;
    const bool more = kt + 1 < nk;
    if (more) TN_GLOAD(kbeg + 32 * (kt + 1));
    compute_stage<MT_W, NT_W>(cur, cur + BM * 64, cur + 2 * BM * 64, cur + 2 * BM * 64 + BN * 64, wm0, wn0, acc, lane,
                              nt_valid);
    if (more) TN_SWRITE(smem + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }
#undef TN_GLOAD
#undef TN_SWRITE

  float* P = partial + (size_t)z * Mout * Nout;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < MT_W; ++i)
#pragma unroll
    for (int j = 0; j < NT_W; ++j) {
      const int col = n0 + wn0 + 32 * j + li;
      if (col >= Nout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < Mout) P[(size_t)row * Nout + col] = acc[i][j][r];
      }
    }
}

// Planes of the logical matrix O[Rp][Cp] (O[r][c] = transpose ? W[c][r] : W[r][c], zero padded) from
// fp32 W[R][C], stored STAGE-MAJOR: element (r, c) at [(c/32)][r][c%32], so that the [Rp][32] B tile
// of one K stage is a single contiguous block (full-line, fully coalesced loads in the NT kernel).
__global__ void split_weight_kernel(const float* __restrict__ W, int R, int C, int transpose, _Float16* hi,
                                    _Float16* lo, int Rp, int Cp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Rp * Cp) return;
  const int kt = i / (Rp * 32), rem = i % (Rp * 32);
  const int r = rem / 32, c = 32 * kt + rem % 32;
  float v = 0.f;
  if (transpose) { if (c < R && r < C) v = W[(size_t)c * C + r]; }
  else { if (r < R && c < C) v = W[(size_t)r * C + c]; }
  const _Float16 h = (_Float16)v;
  hi[i] = h;
  lo[i] = (_Float16)(v - (float)h);
}

// scales[0] = 2^-floor(log2(max|x|)) (1 if the max is 0 or not finite), scales[1] = 1/scales[0]
__global__ void __launch_bounds__(256) amax_partial_kernel(const float* __restrict__ x, int64_t n,
                                                           float* __restrict__ part) {
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = (((uintptr_t)x & 15) == 0) ? n / 4 : 0;
  const f32x4* x4 = (const f32x4*)x;
  for (int64_t i = gid; i < n4; i += stride) {
    const f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  for (int64_t i = 4 * n4 + gid; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(ws[0], ws[1]), fmaxf(ws[2], ws[3]));
}

__global__ void amax_finalize_kernel(const float* __restrict__ part, int nblk, float* scales) {
  float m = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 64) m = fmaxf(m, part[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if (threadIdx.x == 0) {
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)
      s = ldexpf(1.f, 1 - e);        // s*m in [1, 2)
    }
    scales[0] = s;
    scales[1] = 1.f / s;
  }
}

}  // namespace

size_t xgemm_planes_halfs(int Rp, int Cp) { return (size_t)2 * Rp * Cp; }

int launch_split_weight(const float* W, int R, int C, int transpose, void* planes, int Rp, int Cp, hipStream_t st) {
  _Float16* hi = (_Float16*)planes;
  _Float16* lo = hi + (size_t)Rp * Cp;
  const int n = Rp * Cp;
  PROF_LAUNCH("split_weight_kernel", 0.0, 4.0 * R * C + 4.0 * n, st,
              hipLaunchKernelGGL(split_weight_kernel, dim3(cdiv_i(n, 256)), dim3(256), 0, st, W, R, C, transpose, hi,
                                 lo, Rp, Cp));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_amax_scale(const float* x, int64_t n, float* scales, float* part /*>=448 floats*/, hipStream_t st) {
  PROF_LAUNCH("amax_partial_kernel", (double)n, 4.0 * n, st,
              hipLaunchKernelGGL(amax_partial_kernel, dim3(448), dim3(256), 0, st, x, n, part));
  WGNN_CHECK_LAUNCH();
  hipLaunchKernelGGL(amax_finalize_kernel, dim3(1), dim3(64), 0, st, part, 448, scales);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

// C[M][N] = s_out * (s_in*A) B^T + bias; B planes [Np][Kp] with Np = 320 or 448.
int launch_xgemm_nt(const float* A, int lda, int M, int Kp, const void* Bplanes, int Np, float* C, int ldc, int N,
                    const float* bias, const float* s_in, const float* s_out, hipStream_t st) {
  const _Float16* bhi = (const _Float16*)Bplanes;
  const _Float16* blo = bhi + (size_t)Np * Kp;
  const dim3 grid(cdiv_i(M, 128));
  const double fl = 2.0 * M * (double)N * Kp, by = 4.0 * ((double)M * Kp + (double)M * N) + 4.0 * Np * Kp;
  if (Kp % 32 != 0 || lda % 4 != 0) return WGNN_ERR_SHAPE;
  static const int dbg = getenv("WGNN_DBG_NT") ? atoi(getenv("WGNN_DBG_NT")) : 0;   // timing ablations only
#define NT_CASE(NTW)                                                                                              \
  {                                                                                                               \
    const size_t smem = 2 * (size_t)(2 * 128 + 2 * 64 * NTW) * 64;                                                \
    static std::atomic<unsigned long long> done{0};                                                               \
    if (ensure_dyn_smem((const void*)xgemm_nt_kernel<NTW>, smem, done) != WGNN_OK) return WGNN_ERR_HIP;           \
    PROF_LAUNCH("xgemm_nt_kernel<" #NTW ">", fl, by, st,                                                          \
                hipLaunchKernelGGL(xgemm_nt_kernel<NTW>, grid, dim3(XT), smem, st, A, lda, M, Kp, bhi, blo, C, ldc, \
                                   N, bias, s_in, s_out, dbg));                                                             \
  }
  switch (Np) {
    case 64: NT_CASE(1) break;
    case 128: NT_CASE(2) break;
    case 192: NT_CASE(3) break;
    case 256: NT_CASE(4) break;
    case 320: NT_CASE(5) break;
    case 384: NT_CASE(6) break;
    case 448: NT_CASE(7) break;
    default: return WGNN_ERR_UNSUPPORTED;
  }
#undef NT_CASE
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int xgemm_nt_np(int N) { int np = cdiv_i(N, 64) * 64; return np <= 448 ? np : -1; }

// partial[z][Mout][Nout]; M (= A columns) <= 320.
int launch_xgemm_tn(const float* A, int lda, int mcols, const float* B, int ldb, int ncols_b, int ones_col,
                    int shift_T, int K, int splitk, float* partial, int Mout, int Nout, const float* s_in,
                    hipStream_t st) {
  const int kchunk = cdiv_i(cdiv_i(K, splitk), 32) * 32;
  const double fl = 2.0 * Mout * (double)Nout * K;
  const double by = 4.0 * ((double)K * mcols + (double)K * ncols_b + (double)splitk * Mout * Nout);
#define TN_CASE(MTW, NTW)                                                                                          \
  {                                                                                                                \
    const size_t smem = 2 * (size_t)(2 * 64 * MTW + 2 * 128 * NTW) * 64;                                           \
    static std::atomic<unsigned long long> done{0};                                                                \
    if (ensure_dyn_smem((const void*)xgemm_tn_kernel<MTW, NTW>, smem, done) != WGNN_OK) return WGNN_ERR_HIP;       \
    const dim3 grid(splitk, cdiv_i(Nout, 128 * NTW));                                                              \
    PROF_LAUNCH("xgemm_tn_kernel<" #MTW "," #NTW ">", fl, by, st,                                                  \
                hipLaunchKernelGGL((xgemm_tn_kernel<MTW, NTW>), grid, dim3(XT), smem, st, A, lda, mcols, B, ldb,    \
                                   ncols_b, ones_col, shift_T, K, kchunk, partial, Mout, Nout, s_in));           \
  }
  const int mt = cdiv_i(Mout, 64);
  if (mt < 1 || mt > 5) return WGNN_ERR_UNSUPPORTED;
  switch (mt) {
    case 1: TN_CASE(1, 1) break;
    case 2: TN_CASE(2, 1) break;
    case 3: TN_CASE(3, 1) break;
    case 4: TN_CASE(4, 1) break;
    case 5: TN_CASE(5, 1) break;
    default: return WGNN_ERR_UNSUPPORTED;
  }
#undef TN_CASE
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
