// Large-shape instance of the NT plane GEMM: C[M][N] (fp32) (+)= A[M][Kp] . B[N][Kp]^T on fp16 hi / lo
// planes, for products whose M, N and K are all large -- BASELINE configs[4]'s projections (S = 4096 stations, H = 12288:
// GI = g W_ih^T is 3072 x 36 864 x 53 248, dg = dGI W_ih is 3072 x 53 248 x 36 864; reference: the input half of nn.GRU and its
// backward, src/step6_gcn_gru_combined_model.py:23).  pgemm_nt_kernel's 192 x 448 tile of 16x16x32 MFMAs was shaped for the
// 34-station widths (N = 306 / 442: one N slice, A staged once per M tile); at these sizes that argument is gone and what is
// left is an ordinary large GEMM, for which this file has:
//   * a 256 x 256 workgroup tile, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x16_f16 (2.3
//     PFLOP/s sustained against 1.6 for the 16x16x32 form: tools/mfma_rate.hip); 128 accumulator registers per lane;
//   * the same operand formats as pgemm_nt_kernel (A: row-major planes; B: the stage-major, fragment-major image split_weight2
//     writes, common.h bimg_off), staged by LDS-DMA in 1 KB pieces into a two-stage ring (2 x 64 KB); A's XOR swizzle is applied
//     on the source address so that its 32-row fragment reads (ds_read_b128, lane l: row l & 31, 16-byte chunk 2 s + (l >> 5))
//     are conflict-free, B's pieces are read as they lie (k chunk major: conflict-free as well);
//   * three passes per fragment pair (lo*hi + hi*lo + hi*hi), or two when A is a single plane (WGNN_MATH_F16X3G's dg);
//   * tiles dealt so that the M tiles of one N tile run on ONE XCD back to back: the weight tile (the 7.85 GB operand) is
//     fetched from HBM once and re-read from that XCD's L2.
//   * optionally (AIMG) the A operand as an IMAGE too: a pass of its own (repack_a_kernel, 1.3 GB of traffic for g at
//     configs[4] against the GEMM's 184 GB of staging) rewrites the row-major planes as stage-major, fragment-major planes
//     exactly like B's, so that every LDS-DMA piece of either operand is 1 KB contiguous.  Why: a row-major A is staged in
//     64-byte row segments (32 halfs of K per row and stage); those come from beyond L2 here (the 50 MB of one K chunk of g
//     against 4 MB of L2) at a fraction of the rate of whole lines, and the first cut of this kernel, with A row-major, ran
//     exactly as fast as the kernel it was to replace (62.9 vs 60.6 ms per step for GI + dg: profiles/r5_c5_big_gemm.txt).
// K chunks (<= 4096 per launch, chunk sums added onto C) are the caller's, exactly as for pgemm_nt_kernel.
#include <string>

#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BG_BM = 256, BG_BN = 256, BG_WAVES = 8;
constexpr int BG_PLANE = 256 * 64;                               // bytes of one operand plane of one stage (256 rows x 64 B)

// 16-byte chunk c of the 64-byte row `row` sits at chunk c ^ f(row), f = (row >> 3) & 3: a ds_read_b128 is served in four
// 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32 for the upper half-wave: MI355X_MICROARCH.md, LDS); with lane l
// reading row l & 31 the rows of a group are four quads whose (row >> 3) are {0, 1, 2, 3} -- four different 16-byte slots of
// the 64 banks' 256 bytes for each of the four rows mod 4: conflict-free
__device__ __forceinline__ int swz32(int row) { return (row >> 3) & 3; }

// Row-major fp16 planes A[M][lda] -> image planes (bimg_off with Np = Mp rows; rows >= M are zeros).  One wave per 1 KB
// fragment (16 rows x 32 k): lane l reads 16 bytes of row l >> 2 (64 contiguous bytes per row), writes them to chunk l & 3.
__global__ void __launch_bounds__(256) repack_a_kernel(const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo, int lda,
                                                      int M, int Kp, _Float16* __restrict__ out, int Mp, size_t oplane) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const size_t frag = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nfrag = (size_t)(Kp >> 5) * (size_t)(Mp >> 4);
  if (frag >= nfrag) return;
  const int kt = (int)(frag / (size_t)(Mp >> 4)), mt = (int)(frag % (size_t)(Mp >> 4));
  const int r = lane >> 2, c = lane & 3, row = 16 * mt + r;
  const size_t src = (size_t)row * lda + 32 * kt + 8 * c, dst = frag * 512 + (size_t)(c * 128 + r * 8);
  u32x4 vh = {0u, 0u, 0u, 0u}, vl = {0u, 0u, 0u, 0u};
  if (row < M) {
    vh = *(const u32x4*)(Ahi + src);
    if (Alo) vl = *(const u32x4*)(Alo + src);
  }
  *(u32x4*)(out + dst) = vh;
  if (Alo) *(u32x4*)(out + oplane + dst) = vl;
}

template <bool ALO, bool AIMG>
__global__ void __launch_bounds__(64 * BG_WAVES) pgemm_nt256_kernel(const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo,
                                                                   int lda, int M, int Kp, const _Float16* __restrict__ Bpl,
                                                                   int Np, size_t bplane, float* __restrict__ C, int ldc, int N,
                                                                   int nmt, int nnt, int accumulate) {
  constexpr int PLA = ALO ? 2 : 1;
  constexpr int A_SLOT = PLA * BG_PLANE, B_SLOT = 2 * BG_PLANE, STAGE = A_SLOT + B_SLOT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  // blocks b, b + 8, ... share an XCD (round-robin dispatch: speed only): XCD x takes the N tiles x, x + 8, ... and runs the
  // nmt M tiles of each back to back
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int nt = (idx / nmt) * 8 + xcd, mt = idx % nmt;
  if (nt >= nnt) return;
  const int m0 = mt * BG_BM, n0 = nt * BG_BN;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- LDS-DMA: one wave-instruction moves a 1 KB piece = 16 rows x 64 B (lane l: row l >> 2, 16-byte position l & 3); the
  // image is lane-linear, the swizzle is applied to the SOURCE chunk.  Per stage and plane 16 pieces; wave w moves pieces w and
  // w + 8 of every plane: 2 (PLA + 2) loads per wave and stage.
  const int prow = lane >> 2, ppos = lane & 3;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  const _Float16* srcA[PLA][2];
  const _Float16* srcB[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int blk = wave + 8 * q, row = 16 * blk + prow;
    const int chunk = 8 * (ppos ^ swz32(row));
    if (AIMG) {       // A is an image with lda = its padded row count Mp: piece (blk) of this tile is 1 KB, linear
      srcA[0][q] = Ahi + (size_t)(m0 / 16 + blk) * 512 + lane * 8;
      if (ALO) srcA[PLA - 1][q] = Alo + (size_t)(m0 / 16 + blk) * 512 + lane * 8;
    } else {
      const int gr = min(m0 + row, M - 1);                         // rows past M: computed, never stored
      srcA[0][q] = Ahi + (size_t)gr * lda + chunk;
      if (ALO) srcA[PLA - 1][q] = Alo + (size_t)gr * lda + chunk;
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) srcB[pl][q] = Bpl + (size_t)pl * bplane + (size_t)(n0 / 16 + blk) * 512 + lane * 8;   // bimg_off: linear 1 KB
  }
  auto dma = [&](int slot, int kt) {
    char* base = smem + slot * STAGE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int blk = wave + 8 * q;
#pragma unroll
      for (int pl = 0; pl < PLA; ++pl)
        __builtin_amdgcn_global_load_lds((glb_void*)(srcA[pl][q] + (AIMG ? (size_t)lda * 32 * kt : (size_t)32 * kt)),
                                         (lds_void*)(base + pl * BG_PLANE + blk * 1024), 16, 0, 0);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        __builtin_amdgcn_global_load_lds((glb_void*)(srcB[pl][q] + (size_t)Np * 32 * kt),
                                         (lds_void*)(base + A_SLOT + pl * BG_PLANE + blk * 1024), 16, 0, 0);
    }
  };
  // fragment addresses: lane l reads row r = l & 31 of its 32-row tile, chunk 2 s + (l >> 5) of K sub-step s (16 deep)
  const int r32 = lane & 31, hsel = lane >> 5;
  int offA[4][2], offB[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 128 * wm + 32 * i + r32;
      offA[i][s] = AIMG ? (row >> 4) * 1024 + (2 * s + hsel) * 256 + (row & 15) * 16
                        : row * 64 + (((2 * s + hsel) ^ swz32(row)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 64 * wn + 32 * j + r32;       // B pieces are fragment-major: [16-row piece][k chunk][row & 15][16 B]
      offB[j][s] = (row >> 4) * 1024 + (2 * s + hsel) * 256 + (row & 15) * 16;
    }
  }
  auto compute = [&](const char* st) {
    const char* Ah = st;
    const char* Al = st + BG_PLANE;
    const char* Bh = st + A_SLOT;
    const char* Bl = Bh + BG_PLANE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      h8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = *(const h8*)(Bh + offB[j][s]);
        bl[j] = *(const h8*)(Bl + offB[j][s]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[i] = *(const h8*)(Ah + offA[i][s]);
        if (ALO) al[i] = *(const h8*)(Al + offA[i][s]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (ALO) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  const int nk = Kp / 32;
  dma(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // all waves: stage kt has landed, stage kt - 1 is no longer being read
    if (kt + 1 < nk) dma((kt + 1) & 1, kt + 1);
    compute(smem + (kt & 1) * STAGE);
  }

  // ---- epilogue: C tile value (i, j, reg) sits at row 32 i + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), column 32 j + (lane & 31):
  // a half-wave stores 128 contiguous bytes per row
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * wn + 32 * j + r32;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 128 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hsel;
        if (row < M) {
          float* dst = C + (size_t)row * ldc + col;
          *dst = accumulate ? *dst + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

}  // namespace

// Is this product one for the large-shape kernel?  Many rows AND many columns AND a long contraction (configs[4]); the
// 34-station widths (N <= 448) and the few-row per-step products of the wide recurrence keep pgemm_nt_kernel.
bool pgemm_nt256_wanted(int M, int N, int Kp) { return M >= 1024 && N >= 2048 && Kp >= 1024; }

// Bytes of the A-image scratch for an M x Kp operand with `planes` planes (0 when the shape is not the large kernel's).
size_t pgemm_nt256_aimg_bytes(int M, int N, int Kp, int planes) {
  if (!pgemm_nt256_wanted(M, N, Kp)) return 0;
  return (size_t)planes * (size_t)cdiv_i(M, BG_BM) * BG_BM * (size_t)Kp * 2;
}

// Rewrite the row-major planes of A as image planes in `img` (hi plane, then lo plane if Alo): see repack_a_kernel.
int launch_pgemm_repack_a(const void* Ahi, const void* Alo, int lda, int M, int Kp, void* img, hipStream_t st) {
  if (Kp % 32 != 0 || lda % 8 != 0) return WGNN_ERR_SHAPE;
  const int Mp = cdiv_i(M, BG_BM) * BG_BM;
  const size_t oplane = (size_t)Mp * Kp, nfrag = (size_t)(Kp / 32) * (Mp / 16);
  PROF_LAUNCH("repack_a_kernel", 0.0, (Alo ? 2.0 : 1.0) * (2.0 * M * (double)Kp + 2.0 * Mp * (double)Kp), st,
              hipLaunchKernelGGL(repack_a_kernel, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, st, (const _Float16*)Ahi,
                                 (const _Float16*)Alo, lda, M, Kp, (_Float16*)img, Mp, oplane));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

// One K chunk: C (+)= A[:, k0 : k0 + klen] . B[:, k0 : k0 + klen]^T.  Same operand contract as launch_pgemm_nt; Np >= the N
// tiles' rows (pgemm_nt_np(N) is).  a_image: Ahi / Alo are the planes of an image written by launch_pgemm_repack_a (then lda
// is ignored: the image's row count is the padded M).
int launch_pgemm_nt256(const void* Ahi, const void* Alo, int lda, int M, int k0, int klen, const void* Bplanes, int Np,
                       size_t bplane, float* C, int ldc, int N, bool accumulate, hipStream_t st, bool a_image) {
  if (klen % 32 != 0 || k0 % 32 != 0 || lda % 8 != 0) return WGNN_ERR_SHAPE;
  const int nmt = cdiv_i(M, BG_BM), nnt = cdiv_i(N, BG_BN);
  if (Np < nnt * BG_BN) return WGNN_ERR_SHAPE;
  const bool alo = Alo != nullptr;
  const size_t smem = 2 * (size_t)((alo ? 2 : 1) + 2) * BG_PLANE;
  const dim3 grid(8 * nmt * cdiv_i(nnt, 8)), block(64 * BG_WAVES);
  const int Mp = nmt * BG_BM;
  const size_t aoff = a_image ? (size_t)(k0 / 32) * Mp * 32 : (size_t)k0;
  const _Float16* ah = (const _Float16*)Ahi + aoff;
  const _Float16* al = alo ? (const _Float16*)Alo + aoff : ah;
  const int ldk = a_image ? Mp : lda;
  const _Float16* bp = (const _Float16*)Bplanes + (size_t)(k0 / 32) * Np * 32;
  const double fl = 2.0 * M * (double)N * klen;
  const double by = (alo ? 4.0 : 2.0) * (double)M * klen + 4.0 * (double)N * klen + 4.0 * (double)M * N * (accumulate ? 2 : 1);
  static std::atomic<unsigned long long> done[4] = {{0}, {0}, {0}, {0}};
#define BG_GO(ALOV, IMGV, SLOT, NAME)                                                                                   \
  do {                                                                                                                  \
    if (ensure_dyn_smem((const void*)pgemm_nt256_kernel<ALOV, IMGV>, smem, done[SLOT]) != WGNN_OK) return WGNN_ERR_HIP; \
    PROF_LAUNCH(NAME, fl, by, st,                                                                                       \
                hipLaunchKernelGGL((pgemm_nt256_kernel<ALOV, IMGV>), grid, block, smem, st, ah, al, ldk, M, klen, bp, Np, \
                                   bplane, C, ldc, N, nmt, nnt, accumulate ? 1 : 0));                                  \
  } while (0)
  if (alo && a_image) BG_GO(true, true, 0, "pgemm_nt256_kernel");
  else if (alo) BG_GO(true, false, 1, "pgemm_nt256_kernel<rowA>");
  else if (a_image) BG_GO(false, true, 2, "pgemm_nt256_kernel<x2>");
  else BG_GO(false, false, 3, "pgemm_nt256_kernel<x2,rowA>");
#undef BG_GO
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
