// Plane GEMMs: split-fp16 ("f16x3") MFMA GEMMs whose operands already live in HBM as fp16 hi/lo
// planes written by their producers (gcnx_fwd -> g, grux_bwd -> dGI/dGH, grux_fwd -> Y planes,
// split_weight -> W_ih).  No conversion work is left in the GEMMs: staging is a pure 16-byte copy.
//
//   NT  C[M][N] (fp32) = A[M][Kp] . B[N][Kp]^T          GI = g W_ih^T (+b_ih via the ones column),
//                                                       dg = dGI W_ih
//   TN  P[z][Mo][No]   = sum_k A[k][m] * B[k][n]        dW_ih = dGI^T [g|1], dW_hh = dGH^T [Hprev|1]
//
// Both kernels use 4-wave workgroups sized so that TWO workgroups share a CU (<= 80 KB LDS,
// <= 256 VGPRs): with one 8-wave workgroup per CU every wave is in the same phase at the same time
// and load / LDS-write / MFMA / epilogue phases simply add up (measured); two independent
// workgroups overlap them.
//
// NT fragments are K-contiguous rows (ds_read_b128, 64-B rows XOR-swizzled by (row>>2)&3).
// TN operands are K-strided in memory, so tiles are staged row-major [k][cols] exactly as they lie
// in HBM and the MFMA fragments are formed by ds_read_b64_tr_b16 (hardware transpose read); the row
// stride of 320 B (== 64 mod 256) makes the 4 rows x 64 B touched by a half-wave conflict-free.
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PT = 256;   // 4 waves

__device__ __forceinline__ int sw_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ f32x16 mfma_h(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// NT.  BM = 128 (wave w owns rows 32w..32w+31), BN = 32*NT_W columns (one N slice per block).
template <int NT_W, bool X3>
__global__ void __launch_bounds__(PT, 2) pgemm_nt_kernel(const _Float16* __restrict__ Ahi,
                                                         const _Float16* __restrict__ Alo, int lda, int M, int Kp,
                                                         const _Float16* __restrict__ Bpl, int Np,
                                                         float* __restrict__ C, int ldc, int N,
                                                         const float* __restrict__ s_out_p, int nm, int nslices) {
  constexpr int BM = 128, BN = 32 * NT_W, STAGE = (2 * BM + 2 * BN) * 64;
  constexpr int RING = 2;   // 2 x 36 KB stages => two workgroups per CU (measured: beats a 4-deep ring at 1 WG/CU, 111 vs 174 us)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases go to M0
  // blocks {b, b+8, ...} (same XCD, dispatched together) are the N slices of one M tile: the A tile is
  // fetched from HBM once and re-read from that XCD's L2 by the sibling slices.
  const int grp = blockIdx.x / (8 * nslices), within = blockIdx.x % (8 * nslices);
  const int slice = within / 8, mt = grp * 8 + within % 8;
  if (mt >= nm) return;
  const int m0 = mt * BM, n0 = slice * BN;
  const float s_out = s_out_p ? s_out_p[1] : 1.f;

  f32x16 acc[NT_W];
#pragma unroll
  for (int j = 0; j < NT_W; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- LDS-DMA staging (global_load_lds_dwordx4): no VGPRs, no ds_write, and no compiler-inserted
  // vmcnt waits in the way (with register staging hipcc waited for the NEW loads before every
  // ds_write of the OLD ones, so loads never overlapped the MFMAs).  One wave-instruction moves a
  // 1 KB "piece" = 16 tile rows x 64 B; the LDS image is lane-linear, so the XOR swizzle the fragment
  // reads expect is applied to the SOURCE chunk index: LDS (row, pos) <- global chunk pos ^ ((row>>2)&3).
  constexpr int PL = X3 ? 2 : 1;                                        // planes moved: hi (+ lo)
  constexpr int APIECES = PL * BM / 16, BPIECES = PL * BN / 16;        // per stage
  constexpr int NPA = APIECES / 4, NPB = (BPIECES + 3) / 4;            // per wave
  const int prow = lane >> 2, ppos = lane & 3;
  const _Float16* a_src[NPA];
  int a_dst[NPA];
#pragma unroll
  for (int it = 0; it < NPA; ++it) {
    const int p = wave + 4 * it;
    const int plane = p / (BM / 16), row = 16 * (p % (BM / 16)) + prow;
    const int gr = min(m0 + row, M - 1);                 // rows past M are computed but never stored
    a_src[it] = (plane ? Alo : Ahi) + (size_t)gr * lda + 8 * (ppos ^ ((row >> 2) & 3));
    a_dst[it] = plane * BM * 64 + 16 * (p % (BM / 16)) * 64;            // wave-uniform piece base
  }
  const _Float16* b_src[NPB];
  int b_dst[NPB];
  bool b_on[NPB];
#pragma unroll
  for (int it = 0; it < NPB; ++it) {
    const int p = wave + 4 * it;
    b_on[it] = p < BPIECES;                              // wave-uniform
    const int pp = b_on[it] ? p : 0;
    const int plane = pp / (BN / 16), row = 16 * (pp % (BN / 16)) + prow;
    b_src[it] = Bpl + (size_t)plane * Np * Kp + (size_t)(n0 + row) * 32 + 8 * (ppos ^ ((row >> 2) & 3));
    b_dst[it] = 2 * BM * 64 + plane * BN * 64 + 16 * (pp % (BN / 16)) * 64;
  }
  const int nk = Kp / 32;
  const size_t bkt = (size_t)Np * 32;
  const int li = lane & 31, lh = lane >> 5;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;

  auto dma_stage = [&](char* st, int kt) {
#pragma unroll
    for (int it = 0; it < NPA; ++it)
      __builtin_amdgcn_global_load_lds((glb_void*)(a_src[it] + 32 * kt), (lds_void*)(st + a_dst[it]), 16, 0, 0);
#pragma unroll
    for (int it = 0; it < NPB; ++it)
      if (b_on[it])
        __builtin_amdgcn_global_load_lds((glb_void*)(b_src[it] + bkt * kt), (lds_void*)(st + b_dst[it]), 16, 0, 0);
  };
  auto compute = [&](const char* cur) {
    const char* Ah = cur;
    const char* Al = cur + BM * 64;
    const char* Bh = cur + 2 * BM * 64;
    const char* Bl = Bh + BN * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + lh;
      const int aoff = sw_off(32 * wave + li, c);
      const h8 ah = *(const h8*)(Ah + aoff);
      h8 al = ah;
      if (X3) al = *(const h8*)(Al + aoff);
#pragma unroll
      for (int j = 0; j < NT_W; ++j) {
        const int boff = sw_off(32 * j + li, c);
        const h8 bh = *(const h8*)(Bh + boff);
        if (X3) {
          const h8 bl = *(const h8*)(Bl + boff);
          acc[j] = mfma_h(al, bh, acc[j]);
          acc[j] = mfma_h(ah, bl, acc[j]);
        }
        acc[j] = mfma_h(ah, bh, acc[j]);
      }
    }
  };

  // RING-deep LDS ring: stage kt+RING-1 is issued while stage kt is multiplied, so RING-1 stages of
  // DMA are in flight per workgroup (the kernel is bound by memory latency x bytes in flight, not by
  // issue).  vmcnt is counted: (RING-2) stages * (NPA+NPB) DMA instructions may stay outstanding.
  constexpr int PER = NPA + NPB;
  static_assert(RING == 2 || BPIECES % 4 == 0, "counted vmcnt needs the same DMA count in every wave");
#pragma unroll
  for (int s0 = 0; s0 < RING - 1; ++s0)
    if (s0 < nk) dma_stage(smem + s0 * STAGE, s0);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed once at most the DMA of the (up to RING-2) younger stages is outstanding
    const int younger = min(RING - 2, nk - 1 - kt);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // all waves: stage kt visible, stage kt-1 no longer being read
    if (kt + RING - 1 < nk) dma_stage(smem + ((kt + RING - 1) % RING) * STAGE, kt + RING - 1);
    compute(smem + (kt % RING) * STAGE);
  }

#pragma unroll
  for (int j = 0; j < NT_W; ++j) {
    const int col = n0 + 32 * j + li;
    if (col >= N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < M) C[(size_t)row * ldc + col] = acc[j][r] * s_out;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TN.  BM = 32*MT_W columns of A (every wave), BN = 128 columns of B (wave w owns 32w..32w+31).
// grid = (splitk, nMblocks * nNblocks): blocks with the same z and different tiles differ by a
// multiple of splitk in linear id; splitk is a multiple of 8, so they share an XCD's L2.
template <int MT_W, bool X3>
__global__ void __launch_bounds__(PT, 2) pgemm_tn_kernel(const _Float16* __restrict__ Ahi,
                                                         const _Float16* __restrict__ Alo, int lda,
                                                         const _Float16* __restrict__ Bhi,
                                                         const _Float16* __restrict__ Blo, int ldb, int shift_T,
                                                         int K, int kchunk, float* __restrict__ partial, int Mout,
                                                         int Nout, int nNb) {
  constexpr int BM = 32 * MT_W, BN = 128;
  constexpr int RS = 320;                                   // LDS row stride in bytes (== 64 mod 256)
  constexpr int ACH = BM / 8, BCH = BN / 8;                 // 16-byte chunks per tile row
  constexpr int PLANE = 32 * RS, STAGE = 4 * PLANE;         // Ahi, Alo, Bhi, Blo
  constexpr int PL = X3 ? 2 : 1;                           // planes staged: hi (+ lo)
  constexpr int NA = (PL * 32 * ACH + PT - 1) / PT, NB = (PL * 32 * BCH + PT - 1) / PT;
  static_assert(BM * 2 <= RS && BN * 2 <= RS, "tile rows must fit the LDS row stride");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.x;
  const int mb = blockIdx.y / nNb, nb = blockIdx.y % nNb;
  const int m0 = mb * BM, n0 = nb * BN;
  const int kbeg = z * kchunk, kend = min(K, kbeg + kchunk);

  f32x16 acc[MT_W];
#pragma unroll
  for (int i = 0; i < MT_W; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // staging maps: chunk q -> (plane, k row, 16-B column chunk)
  int a_row[NA], a_col[NA], a_dst[NA];
  bool a_on[NA], a_lo[NA];
#pragma unroll
  for (int it = 0; it < NA; ++it) {
    const int q = tid + PT * it;
    a_on[it] = q < PL * 32 * ACH;
    const int qq = a_on[it] ? q : 0;
    const int plane = qq / (32 * ACH), rem = qq % (32 * ACH);
    a_lo[it] = plane != 0;
    a_row[it] = rem / ACH;
    a_col[it] = m0 + 8 * (rem % ACH);
    a_on[it] = a_on[it] && a_col[it] < lda;                 // planes are padded to multiples of 8 columns
    a_dst[it] = plane * PLANE + a_row[it] * RS + 16 * (rem % ACH);
  }
  int b_row[NB], b_col[NB], b_dst[NB];
  bool b_on[NB], b_lo[NB];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    const int q = tid + PT * it;
    b_on[it] = q < PL * 32 * BCH;
    const int qq = b_on[it] ? q : 0;
    const int plane = qq / (32 * BCH), rem = qq % (32 * BCH);
    b_lo[it] = plane != 0;
    b_row[it] = rem / BCH;
    b_col[it] = n0 + 8 * (rem % BCH);
    b_on[it] = b_on[it] && b_col[it] < ldb;
    b_dst[it] = (2 + plane) * PLANE + b_row[it] * RS + 16 * (rem % BCH);
  }
  u32x4 ra[NA], rb[NB];
  const u32x4 zero = {0u, 0u, 0u, 0u};

#define TN_LOAD(k0)                                                                                  \
  do {                                                                                               \
    _Pragma("unroll") for (int it = 0; it < NA; ++it) {                                              \
      const int k = (k0) + a_row[it];                                                                \
      const bool ok = a_on[it] && k < kend;                                                          \
      const u32x4 v = *(const u32x4*)((a_lo[it] ? Alo : Ahi) + (size_t)(ok ? k : kbeg) * lda + (ok ? a_col[it] : 0)); \
      ra[it] = ok ? v : zero;                                                                        \
    }                                                                                                \
    _Pragma("unroll") for (int it = 0; it < NB; ++it) {                                              \
      const int k = (k0) + b_row[it];                                                                \
      const bool ok = b_on[it] && k < kend;                                                          \
      int kr = k;                                                                                    \
      if (shift_T > 0) kr = (k % shift_T) != 0 ? k - 1 : K;   /* row K = the stored t = 0 row */     \
      const u32x4 v = *(const u32x4*)((b_lo[it] ? Blo : Bhi) + (size_t)(ok ? kr : kbeg) * ldb + (ok ? b_col[it] : 0)); \
      rb[it] = ok ? v : zero;                                                                        \
    }                                                                                                \
  } while (0)
#define TN_STORE(st)                                                                                 \
  do {                                                                                               \
    _Pragma("unroll") for (int it = 0; it < NA; ++it) if (tid + PT * it < PL * 32 * ACH) *(u32x4*)((st) + a_dst[it]) = ra[it]; \
    _Pragma("unroll") for (int it = 0; it < NB; ++it) if (tid + PT * it < PL * 32 * BCH) *(u32x4*)((st) + b_dst[it]) = rb[it]; \
  } while (0)

  const int nk = (kend - kbeg + 31) / 32;
  if (nk > 0) {
    TN_LOAD(kbeg);
    TN_STORE(smem);
  }
  __syncthreads();
  // transpose-read addressing: 16-lane group x = lane>>4 handles 16 columns (x&1) and k-half (x>>1);
  // inside a group lane 4q+p supplies the address of row q, columns 4p..4p+3 (8 bytes).
  const int x = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int tr_base = (8 * (x >> 1) + q4) * RS + (16 * (x & 1) + 4 * p4) * 2;
  typedef __attribute__((address_space(3))) fh4 lds_fh4;
  auto trread = [&](const char* tile, int koff, int coff) -> h8 {
    const char* p0 = tile + tr_base + koff * RS + coff * 2;
    const fh4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4*)p0);
    const fh4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4*)(p0 + 4 * RS));
    const h4 w0 = __builtin_bit_cast(h4, v0), w1 = __builtin_bit_cast(h4, v1);
    h8 r;
    r[0] = w0[0]; r[1] = w0[1]; r[2] = w0[2]; r[3] = w0[3];
    r[4] = w1[0]; r[5] = w1[1]; r[6] = w1[2]; r[7] = w1[3];
    return r;
  };
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * STAGE;
    const bool more = kt + 1 < nk;
    if (more) TN_LOAD(kbeg + 32 * (kt + 1));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const h8 bh = trread(cur + 2 * PLANE, 16 * ks, 32 * wave);
      h8 bl = bh;
      if (X3) bl = trread(cur + 3 * PLANE, 16 * ks, 32 * wave);
#pragma unroll
      for (int i = 0; i < MT_W; ++i) {
        const h8 ah = trread(cur, 16 * ks, 32 * i);
        if (X3) {
          const h8 al = trread(cur + PLANE, 16 * ks, 32 * i);
          acc[i] = mfma_h(al, bh, acc[i]);
          acc[i] = mfma_h(ah, bl, acc[i]);
        }
        acc[i] = mfma_h(ah, bh, acc[i]);
      }
    }
    if (more) TN_STORE(smem + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }
#undef TN_LOAD
#undef TN_STORE

  float* P = partial + (size_t)z * Mout * Nout;
  const int li = lane & 31, lh = lane >> 5;
  const int col = n0 + 32 * wave + li;
  if (col < Nout) {
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < Mout) P[(size_t)row * Nout + col] = acc[i][r];
      }
  }
}

// Planes of O[Rp][Cp] (O[r][c] = transpose ? W[c][r] : W[r][c]; optional extra column `bias_col`
// holding bias[r]; zero elsewhere) from fp32 W[R][C], STAGE-MAJOR: (r,c) at [(c/32)][r][c%32].
__global__ void split_weight2_kernel(const float* __restrict__ W, int R, int C, int transpose,
                                     const float* __restrict__ bias, int bias_col, _Float16* hi, _Float16* lo,
                                     int Rp, int Cp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Rp * Cp) return;
  const int kt = i / (Rp * 32), rem = i % (Rp * 32);
  const int r = rem / 32, c = 32 * kt + rem % 32;
  float v = 0.f;
  if (transpose) { if (c < R && r < C) v = W[(size_t)c * C + r]; }
  else {
    if (r < R && c < C) v = W[(size_t)r * C + c];
    else if (bias && r < R && c == bias_col) v = bias[r];
  }
  const _Float16 h = (_Float16)v;
  hi[i] = h;
  lo[i] = (_Float16)(v - (float)h);
}

}  // namespace

int launch_split_weight2(const float* W, int R, int C, int transpose, const float* bias, int bias_col, void* planes,
                         int Rp, int Cp, hipStream_t st) {
  _Float16* hi = (_Float16*)planes;
  _Float16* lo = hi + (size_t)Rp * Cp;
  const int n = Rp * Cp;
  PROF_LAUNCH("split_weight2_kernel", 0.0, 4.0 * R * C + 4.0 * n, st,
              hipLaunchKernelGGL(split_weight2_kernel, dim3(cdiv_i(n, 256)), dim3(256), 0, st, W, R, C, transpose,
                                 bias, bias_col, hi, lo, Rp, Cp));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int pgemm_nt_np(int N) { return cdiv_i(N, 160) * 160; }

// C[M][N] = s_out * A B^T.  A planes [M][lda] (Kp <= lda), B stage-major planes with Np = pgemm_nt_np(N) rows.
int launch_pgemm_nt(const void* Ahi, const void* Alo, int lda, int M, int Kp, const void* Bplanes, int Np, float* C,
                    int ldc, int N, const float* s_out, bool x3, hipStream_t st) {
  if (Kp % 32 != 0 || lda % 8 != 0 || Np % 160 != 0) return WGNN_ERR_SHAPE;
  constexpr int NTW = 5;
  const int nm = cdiv_i(M, 128), nslices = Np / 160;
  const int grid = cdiv_i(nm, 8) * 8 * nslices;
  const size_t smem = 2 * (size_t)(2 * 128 + 2 * 32 * NTW) * 64;
  static std::atomic<unsigned long long> done{0}, done16{0};
  if (ensure_dyn_smem((const void*)pgemm_nt_kernel<NTW, true>, smem, done) != WGNN_OK) return WGNN_ERR_HIP;
  if (ensure_dyn_smem((const void*)pgemm_nt_kernel<NTW, false>, smem, done16) != WGNN_OK) return WGNN_ERR_HIP;
  const double fl = 2.0 * M * (double)N * Kp;
  const double by = (x3 ? 4.0 : 2.0) * ((double)M * Kp + (double)Np * Kp) + 4.0 * (double)M * N;
  if (x3)
    PROF_LAUNCH("pgemm_nt_kernel<5>", fl, by, st,
                hipLaunchKernelGGL((pgemm_nt_kernel<NTW, true>), dim3(grid), dim3(PT), smem, st, (const _Float16*)Ahi,
                                   (const _Float16*)Alo, lda, M, Kp, (const _Float16*)Bplanes, Np, C, ldc, N, s_out, nm,
                                   nslices));
  else
    PROF_LAUNCH("pgemm_nt_kernel<5,f16>", fl, by, st,
                hipLaunchKernelGGL((pgemm_nt_kernel<NTW, false>), dim3(grid), dim3(PT), smem, st, (const _Float16*)Ahi,
                                   (const _Float16*)Alo, lda, M, Kp, (const _Float16*)Bplanes, Np, C, ldc, N, s_out, nm,
                                   nslices));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

// partial[z][Mout][Nout] = sum over k chunk z of A[k][m] B[k][n].  A planes [K][lda], B planes [K][ldb].
// shift_T > 0: B row k is taken from row k-1, and from the extra row K (which the producer fills with what the
// operand looks like at a window start) where k % shift_T == 0.
int launch_pgemm_tn(const void* Ahi, const void* Alo, int lda, const void* Bhi, const void* Blo, int ldb, int shift_T,
                    int K, int splitk, float* partial, int Mout, int Nout, bool x3, hipStream_t st) {
  if (lda % 8 != 0 || ldb % 8 != 0) return WGNN_ERR_SHAPE;
  constexpr int MTW = 5;
  const int nMb = cdiv_i(Mout, 32 * MTW), nNb = cdiv_i(Nout, 128);
  const int kchunk = cdiv_i(cdiv_i(K, splitk), 32) * 32;
  const size_t smem = 2 * 4 * 32 * 320;
  static std::atomic<unsigned long long> done{0}, done16{0};
  if (ensure_dyn_smem((const void*)pgemm_tn_kernel<MTW, true>, smem, done) != WGNN_OK) return WGNN_ERR_HIP;
  if (ensure_dyn_smem((const void*)pgemm_tn_kernel<MTW, false>, smem, done16) != WGNN_OK) return WGNN_ERR_HIP;
  const double fl = 2.0 * Mout * (double)Nout * K;
  const double by = (x3 ? 4.0 : 2.0) * ((double)K * Mout + (double)K * Nout) + 4.0 * (double)splitk * Mout * Nout;
  if (x3)
    PROF_LAUNCH("pgemm_tn_kernel<5>", fl, by, st,
                hipLaunchKernelGGL((pgemm_tn_kernel<MTW, true>), dim3(splitk, nMb * nNb), dim3(PT), smem, st,
                                   (const _Float16*)Ahi, (const _Float16*)Alo, lda, (const _Float16*)Bhi,
                                   (const _Float16*)Blo, ldb, shift_T, K, kchunk, partial, Mout, Nout, nNb));
  else
    PROF_LAUNCH("pgemm_tn_kernel<5,f16>", fl, by, st,
                hipLaunchKernelGGL((pgemm_tn_kernel<MTW, false>), dim3(splitk, nMb * nNb), dim3(PT), smem, st,
                                   (const _Float16*)Ahi, (const _Float16*)Alo, lda, (const _Float16*)Bhi,
                                   (const _Float16*)Blo, ldb, shift_T, K, kchunk, partial, Mout, Nout, nNb));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
