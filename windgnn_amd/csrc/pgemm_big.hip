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

template <bool ALO>
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
    const int gr = min(m0 + row, M - 1);                           // rows past M: computed, never stored
    srcA[0][q] = Ahi + (size_t)gr * lda + chunk;
    if (ALO) srcA[PLA - 1][q] = Alo + (size_t)gr * lda + chunk;
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
        __builtin_amdgcn_global_load_lds((glb_void*)(srcA[pl][q] + (size_t)32 * kt), (lds_void*)(base + pl * BG_PLANE + blk * 1024),
                                         16, 0, 0);
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
      offA[i][s] = row * 64 + (((2 * s + hsel) ^ swz32(row)) << 4);
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

// One K chunk: C (+)= A[:, k0 : k0 + klen] . B[:, k0 : k0 + klen]^T.  Same operand contract as launch_pgemm_nt; Np >= the N
// tiles' rows (pgemm_nt_np(N) is).
int launch_pgemm_nt256(const void* Ahi, const void* Alo, int lda, int M, int k0, int klen, const void* Bplanes, int Np,
                       size_t bplane, float* C, int ldc, int N, bool accumulate, hipStream_t st) {
  if (klen % 32 != 0 || k0 % 32 != 0 || lda % 8 != 0) return WGNN_ERR_SHAPE;
  const int nmt = cdiv_i(M, BG_BM), nnt = cdiv_i(N, BG_BN);
  if (Np < nnt * BG_BN) return WGNN_ERR_SHAPE;
  const bool alo = Alo != nullptr;
  const size_t smem = 2 * (size_t)((alo ? 2 : 1) + 2) * BG_PLANE;
  const dim3 grid(8 * nmt * cdiv_i(nnt, 8)), block(64 * BG_WAVES);
  const _Float16* ah = (const _Float16*)Ahi + k0;
  const _Float16* al = alo ? (const _Float16*)Alo + k0 : ah;
  const _Float16* bp = (const _Float16*)Bplanes + (size_t)(k0 / 32) * Np * 32;
  const double fl = 2.0 * M * (double)N * klen;
  const double by = (alo ? 4.0 : 2.0) * (double)M * klen + 4.0 * (double)N * klen + 4.0 * (double)M * N * (accumulate ? 2 : 1);
  static std::atomic<unsigned long long> done3{0}, done2{0};
  if (alo) {
    if (ensure_dyn_smem((const void*)pgemm_nt256_kernel<true>, smem, done3) != WGNN_OK) return WGNN_ERR_HIP;
    PROF_LAUNCH("pgemm_nt256_kernel", fl, by, st,
                hipLaunchKernelGGL((pgemm_nt256_kernel<true>), grid, block, smem, st, ah, al, lda, M, klen, bp, Np, bplane, C, ldc, N,
                                   nmt, nnt, accumulate ? 1 : 0));
  } else {
    if (ensure_dyn_smem((const void*)pgemm_nt256_kernel<false>, smem, done2) != WGNN_OK) return WGNN_ERR_HIP;
    PROF_LAUNCH("pgemm_nt256_kernel<x2>", fl, by, st,
                hipLaunchKernelGGL((pgemm_nt256_kernel<false>), grid, block, smem, st, ah, al, lda, M, klen, bp, Np, bplane, C, ldc, N,
                                   nmt, nnt, accumulate ? 1 : 0));
  }
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
