// Exact-fp32 GEMMs (WGNN_MATH_F32) for B*T >= 4096: the GRU input projection, its
// backward and the weight gradients on v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 with the structure of the plane
// GEMMs (pgemm.hip): one large workgroup per CU, operands staged by LDS-DMA (global_load_lds_dwordx4) exactly as they lie in HBM into a two-stage ring,
// fragments read with wide LDS loads, no per-element bounds code in the main loop.  That needs operands whose rows are
// 16-byte aligned and whose K extent is padded with finite values: g [B*T][Ip] (ones column at I, zeros after: written by
// gcn32_fwd), dGI [B*T][Gp] (zero padding written by the recurrences' backward) and zero-padded copies of W_ih / W_ih^T
// (pad_weight_kernel below; a caller that keeps wgnn_params.prepared never runs it).  fp32 MFMA sustains 149 TFLOP/s here (tools/mfma_rate.hip);
// the register-staged general kernel of gemm.hip (kept for small B*T and the wide-GRU / CSR paths) reaches 72.
//
//   NT  C[M][N] = A[M][Kp] . Bp[N][Kp]^T      GI = [g|1] [W_ih|b_ih]^T,   dg = dGI (W_ih^T)^T
//   TN  P[z][Mo][No] = sum_k A[k][m] B[k][n]  dW_ih|db_ih = dGI^T [g|1], dW_hh|db_hh = dGH^T [Hprev|1]
//                                              (split-K, reduced by finish.hip)
#include <string>
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// ------------------------------------------------------------------------------------------------
// NT.  MW (M) x NW (N) waves; wave tile 32 x 32 T32, workgroup tile 32 MW x 32 T32 NW; K step 32 = one 128-byte LDS row.
// Two forms: 4 x 2 waves with T32 <= 7 (128-row tiles, one workgroup per CU: from 192 such tiles) and 1 x NW waves with
// T32 = 1 (32-row tiles, one wave per 32 output columns: a few thousand rows still give most CUs a workgroup).
// A k group of 8 = two 16-byte chunks: lane (i, kh) of the 32x32x2 MFMA reads chunk 2 kk + kh of row i as ONE
// ds_read_b128; its element jj is the operand of MFMA jj, whose two k slots are therefore k = 8 kk + jj and 8 kk + 4 + jj
// -- any bijection works as long as A and B agree.  The 16-byte chunks of a row are XOR-swizzled by (row >> 1) & 7
// (applied on the DMA source address), which makes the 16 rows x 16 B of a quarter-wave hit 16 different bank groups.
constexpr int G_WAVES = 8;      // the TN kernel's and the big NT form's

template <int MW, int NW, int T32>
__global__ void __launch_bounds__(64 * MW * NW) gemm32_nt_kernel(const float* __restrict__ A, int lda, int M, int Kp,
                                                               const float* __restrict__ Bp, int nb_rows,
                                                               float* __restrict__ C, int ldc, int N, int nm, int nsl,
                                                               int stagger) {
  constexpr int G_BM = 32 * MW, BN = 32 * T32 * NW, NWAVES = MW * NW;
  constexpr int A_BYTES = G_BM * 128, STAGE = A_BYTES + BN * 128;
  constexpr int AP = G_BM / 8, BP = BN / 8;                      // 1 KB pieces: 8 rows x 128 B
  constexpr int NPIECE = AP + BP, NIT = (NPIECE + NWAVES - 1) / NWAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % MW, wn = wave / MW;
  // blocks {b, b + 8, ...} share an XCD: the nsl column slices of one row tile are consecutive there, so the tile's rows of A
  // cross HBM -> L2 once (nsl = 1: the identity)
  const int bq = blockIdx.x / 8, mt = (bq / nsl) * 8 + blockIdx.x % 8, sl = bq % nsl;
  if (mt >= nm) return;
  const int m0 = mt * G_BM, n0 = sl * BN;
  // column tiles of this wave that hold columns of the product (the last slice may be narrower than BN: its dead tiles are
  // staged from clamped rows and never multiplied)
  const int tl = NW == 1 ? min(T32, (((N + 31) >> 5) - sl * T32)) : T32;   // T32 or T32 - 1 (launcher)

  f32x16 acc[T32];
#pragma unroll
  for (int j = 0; j < T32; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const float* src[NIT];
  int dst[NIT];
  bool on[NIT];
  {
    const int r8 = lane >> 3, pos = lane & 7;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int p = wave + NWAVES * it;
      on[it] = p < NPIECE;                                        // wave-uniform
      const int pp = on[it] ? p : 0;
      const bool isA = pp < AP;
      const int blk = isA ? pp : pp - AP;
      const int row = 8 * blk + r8;
      const int chunk = pos ^ ((row >> 1) & 7);
      if (isA) src[it] = A + (size_t)min(m0 + row, M - 1) * lda + 4 * chunk;   // rows past M: computed, never stored
      else src[it] = Bp + (size_t)min(n0 + row, nb_rows - 1) * Kp + 4 * chunk;
      dst[it] = (isA ? 0 : A_BYTES) + blk * 1024;
    }
  }
  auto dma_stage = [&](char* st, int kt) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (on[it]) __builtin_amdgcn_global_load_lds((glb_void*)(src[it] + 32 * kt), (lds_void*)(st + dst[it]), 16, 0, 0);
  };

  const int li = lane & 31, kh = lane >> 5, sw = (li >> 1) & 7;
  const int a_row = (32 * wm + li) * 128, b_row = A_BYTES + (32 * T32 * wn + li) * 128;
  // TL: column tiles multiplied (T32, or T32 - 1 in the narrower last slice of the 4-wave form)
  auto compute = [&](const char* cur, auto tl_c) {
    constexpr int TL = decltype(tl_c)::value;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int co = ((2 * kk + kh) ^ sw) << 4;
      const f32x4 a = *(const f32x4*)(cur + a_row + co);
      f32x4 b[TL];
#pragma unroll
      for (int j = 0; j < TL; ++j) b[j] = *(const f32x4*)(cur + b_row + j * 32 * 128 + co);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int j = 0; j < TL; ++j) acc[j] = mfma32(a[jj], b[j][jj], acc[j]);
    }
  };

#ifndef G32_EPI_FUSED
#define G32_EPI_FUSED 1
#endif
#ifndef G32_ABLATE          // measurement builds only (tools/exp/gemm32_ablate.sh): 1 no C stores, 2 no DMA after the first
#define G32_ABLATE 0        // stage, 4 no barrier / wait either (the LDS-read + MFMA loop alone)
#endif
  const int nk = Kp / 32;
  dma_stage(smem, 0);
  // Two workgroups share a CU in the 4-wave form; the second one of each pair (the dispatcher fills every CU of an XCD once
  // before it comes back: blocks 256 ... 511 of the first 512) starts `stagger` x 3.4 us late, so that from then on one
  // workgroup's epilogue and first stage fall into the other's K loop -- and the chip's C stores do not come as one burst.
  if (stagger > 0 && ((blockIdx.x >> 8) & 1) && blockIdx.x < 512)
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  // The big form finishes its LAST K step column tile by column tile and stores tile j - 1 between the MFMAs of tile j: the
  // 16 store instructions per column tile issue in the shadow of 64-cycle MFMAs instead of after the loop with the matrix pipe
  // idle (the epilogue's 16-18 us per product were store ISSUE time: neither a persistent loop nor a second workgroup hid
  // them, profiles/r5_gemm32_nt_ablation.txt).  Per accumulator the order of its MFMAs is unchanged: bit-identical.
  constexpr bool FUSE_EPI = MW == 4 && NW == 2 && G32_ABLATE == 0 && G32_EPI_FUSED;
  const int nloop = FUSE_EPI ? nk - 1 : nk;
  for (int kt = 0; kt < nloop; ++kt) {
    if (!(G32_ABLATE & 4)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();        // stage kt visible to all waves, stage kt-1 no longer being read
    }
    if (!(G32_ABLATE & 2) && kt + 1 < nk) dma_stage(smem + ((kt + 1) & 1) * STAGE, kt + 1);
    if (T32 == 1 || tl == T32) compute(smem + (kt & 1) * STAGE, std::integral_constant<int, T32>{});
    else compute(smem + (kt & 1) * STAGE, std::integral_constant<int, (T32 > 1 ? T32 - 1 : 1)>{});
  }
  if (FUSE_EPI) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* cur = smem + ((nk - 1) & 1) * STAGE;
    f32x4 a4[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) a4[kk] = *(const f32x4*)(cur + a_row + (((2 * kk + kh) ^ sw) << 4));
    auto store1 = [&](int j, int r) {
      const int col = n0 + 32 * T32 * wn + 32 * j + li;
      const int row = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (col < N && row < M) C[(size_t)row * ldc + col] = acc[j][r];
    };
#pragma unroll
    for (int j = 0; j < T32; ++j) {
      f32x4 b4[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) b4[kk] = *(const f32x4*)(cur + b_row + j * 32 * 128 + (((2 * kk + kh) ^ sw) << 4));
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          acc[j] = mfma32(a4[kk][jj], b4[kk][jj], acc[j]);
          if (j > 0) store1(j - 1, 4 * kk + jj);              // one store of the finished tile per MFMA of this one
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) store1(T32 - 1, r);
    return;
  }
  if (G32_ABLATE & 1) {      // keep the accumulators alive without the stores
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < T32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[j][r];
    if (t == 1234.5678f) C[0] = t;
    return;
  }

#ifndef G32_STORE
#define G32_STORE 0
#endif
  if (G32_STORE & 1) {       // row-major order: a row's T32 x 128 B go out back to back
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
      for (int j = 0; j < T32; ++j) {
        const int col = n0 + 32 * T32 * wn + 32 * j + li;
        if (col < N && row < M) {
          if (G32_STORE & 2) __builtin_nontemporal_store(acc[j][r], C + (size_t)row * ldc + col);
          else C[(size_t)row * ldc + col] = acc[j][r];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < T32; ++j) {
    const int col = n0 + 32 * T32 * wn + 32 * j + li;
    if (col >= N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (row < M) {
        if (G32_STORE & 2) __builtin_nontemporal_store(acc[j][r], C + (size_t)row * ldc + col);
        else C[(size_t)row * ldc + col] = acc[j][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// NT, persistent form of the 4 x 2-wave kernel (one slice of N: nsl = 1).  A workgroup walks its row tiles mt = blockIdx.x,
// + gridDim.x, ... without leaving the CU: the first stage of the NEXT tile is requested during the last K step of the current
// one (into the ring slot that step no longer needs), the tile's C stores are issued behind that request, and the next tile's
// first K step waits with a COUNTED s_waitcnt -- vmcnt counts loads and stores alike and retires them in issue order on
// gfx9, so "at most as many operations outstanding as stores were issued after the stage's loads" means "the stage has
// landed" while the stores are still draining under the K step's MFMAs.  (The non-persistent form pays a workgroup turn-around
// plus a first-stage latency per tile, and its epilogue's drain before s_endpgm: profiles/r5_gemm32_nt_ablation.txt.)
// Same products in the same order per element as gemm32_nt_kernel<4, 2, T32>: bit-identical results.
template <int T32>
__global__ void __launch_bounds__(64 * G_WAVES) gemm32_ntp_kernel(const float* __restrict__ A, int lda, int M, int Kp,
                                                                const float* __restrict__ Bp, int nb_rows,
                                                                float* __restrict__ C, int ldc, int N, int nm) {
  constexpr int G_BM = 128, BN = 64 * T32;
  constexpr int A_BYTES = G_BM * 128, STAGE = A_BYTES + BN * 128;
  constexpr int AP = G_BM / 8, BP = BN / 8;                      // 1 KB pieces; AP = 16 = two per wave, then BP / 8 per wave
  constexpr int NITB = BP / G_WAVES;                             // = T32
  static_assert(AP == 2 * G_WAVES && BP % G_WAVES == 0, "every wave requests the same number of pieces per stage");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  int mt = blockIdx.x;
  if (mt >= nm) return;

  const int r8 = lane >> 3, pos = lane & 7;
  const float* srcB[NITB];
  int dstB[NITB];
#pragma unroll
  for (int it = 0; it < NITB; ++it) {
    const int blk = wave + G_WAVES * it, row = 8 * blk + r8, chunk = pos ^ ((row >> 1) & 7);
    srcB[it] = Bp + (size_t)min(row, nb_rows - 1) * Kp + 4 * chunk;
    dstB[it] = A_BYTES + blk * 1024;
  }
  const float* srcA[2];
  auto set_a = [&](int t) {                                       // the two A pieces of this wave for row tile t
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int blk = wave + G_WAVES * it, row = 8 * blk + r8, chunk = pos ^ ((row >> 1) & 7);
      srcA[it] = A + (size_t)min(t * G_BM + row, M - 1) * lda + 4 * chunk;      // rows past M: computed, never stored
    }
  };
  auto dma_stage = [&](char* st, int kt) {
#pragma unroll
    for (int it = 0; it < 2; ++it)
      __builtin_amdgcn_global_load_lds((glb_void*)(srcA[it] + 32 * kt), (lds_void*)(st + (wave + G_WAVES * it) * 1024), 16, 0, 0);
#pragma unroll
    for (int it = 0; it < NITB; ++it)
      __builtin_amdgcn_global_load_lds((glb_void*)(srcB[it] + 32 * kt), (lds_void*)(st + dstB[it]), 16, 0, 0);
  };

  const int li = lane & 31, kh = lane >> 5, sw = (li >> 1) & 7;
  const int a_row = (32 * wm + li) * 128, b_row = A_BYTES + (32 * T32 * wn + li) * 128;
  int livej = 0;                                                  // column tiles of this wave that hold columns of C (uniform)
#pragma unroll
  for (int j = 0; j < T32; ++j) livej += (32 * T32 * wn + 32 * j < N) ? 1 : 0;
  f32x16 acc[T32];
  const int nk = Kp / 32;
  int slot = 0;
  bool first = true;
  set_a(mt);
  dma_stage(smem, 0);
  for (;;) {
    const int mnext = mt + (int)gridDim.x;
    const bool has_next = mnext < nm;
#pragma unroll
    for (int j = 0; j < T32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt == 0 && !first) {
        // issued since this stage's loads: 16 stores per live column tile of the previous row tile (a tile that has a successor is
        // never the ragged last one: every row < M, every store instruction issued)
        if (livej >= 4) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
        else if (livej == 3) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        else if (livej == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (livej == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();        // this stage visible to all waves, the other slot no longer being read
      char* cur = smem + slot * STAGE;
      char* nxt = smem + (slot ^ 1) * STAGE;
      if (kt + 1 < nk) {
        dma_stage(nxt, kt + 1);
      } else if (has_next) {
        set_a(mnext);
        dma_stage(nxt, 0);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int co = ((2 * kk + kh) ^ sw) << 4;
        const f32x4 a = *(const f32x4*)(cur + a_row + co);
        f32x4 b[T32];
#pragma unroll
        for (int j = 0; j < T32; ++j) b[j] = *(const f32x4*)(cur + b_row + j * 32 * 128 + co);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int j = 0; j < T32; ++j) acc[j] = mfma32(a[jj], b[j][jj], acc[j]);
      }
      slot ^= 1;
    }
    asm volatile("" ::: "memory");          // the stores stay BEHIND the next tile's first-stage loads (the counted wait relies on it)
    const int m0 = mt * G_BM;
#pragma unroll
    for (int j = 0; j < T32; ++j) {
      if (32 * T32 * wn + 32 * j >= N) continue;                  // uniform: a dead column tile issues nothing
      const int col = 32 * T32 * wn + 32 * j + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (has_next) {                                           // not the ragged tile: 16 store instructions, no row test
          if (col < N) C[(size_t)row * ldc + col] = acc[j][r];
        } else if (col < N && row < M) {
          C[(size_t)row * ldc + col] = acc[j][r];
        }
      }
    }
    asm volatile("" ::: "memory");
    if (!has_next) break;
    mt = mnext;
    first = false;
  }
}

// ------------------------------------------------------------------------------------------------
// TN (split-K).  8 waves = 4 (M) x 2 (N); wave tile 80 x 16 T16 of v_mfma_f32_16x16x4_f32, workgroup tile 320 x 32 T16 of
// one K chunk.  Both operands are K-strided in memory: a stage (32 k rows) is staged row-major as it lies in HBM and
// lane (i, kq) of the MFMA reads element i of row 4 kk + kq directly (ds_read_b32).  The LDS row pitch is the tile
// width + 16 floats, i.e. 16 or 48 (mod 64): the four rows one read touches then fall into four different 16-bank
// windows.  The DMA image is linear, so the 16 pad floats of a row are written too (with the row's first chunk).
constexpr int T_BM = 320, T_BK = 32;

// A2 != null: GEMM columns m >= msplit of the A operand are column m - msplit of A2 (row stride lda2; msplit a multiple of
// 4, so no 16-byte chunk straddles the two sources).  Used for dW_hh = [dGI_r | dGI_z | dGH_n]^T [Hprev|1]: the BPTT kernel
// stores the n third of dGH only (its r and z thirds equal dGI's).
template <int T16>
__global__ void __launch_bounds__(64 * G_WAVES) gemm32_tn_kernel(const float* __restrict__ A, int lda,
                                                               const float* __restrict__ B, int ldb, int K, int kchunk,
                                                               int splitk, float* __restrict__ P, int Mo, int No,
                                                               int nNb, int ntiles, const float* __restrict__ A2,
                                                               int lda2, int msplit) {
  constexpr int WN = 16 * T16, BN = 2 * WN;
  constexpr int PA = T_BM + 16, PB = BN + 16;                     // LDS row pitches (floats)
  constexpr int A_BYTES = T_BK * PA * 4, STAGE = A_BYTES + T_BK * PB * 4;
  constexpr int AP = PA / 8, BP = PB / 8;                         // 1 KB pieces
  constexpr int NPIECE = AP + BP, NIT = (NPIECE + G_WAVES - 1) / G_WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  // blocks {b, b+8, ...} share an XCD: they are the tiles of ONE K chunk, so the chunk's rows cross HBM -> L2 once
  const int id = blockIdx.x;
  const int z = (id / (8 * ntiles)) * 8 + id % 8, tile = (id / 8) % ntiles;
  if (z >= splitk) return;
  const int m0 = (tile / nNb) * T_BM, n0 = (tile % nNb) * BN;
  const int kbeg = z * kchunk, kend = min(K, kbeg + kchunk);

  f32x4 acc[5][T16];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < T16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA plan: piece p of the stage image = bytes [1024 p, 1024 p + 1024); lane -> (row, column) through the pitch
  int roff[NIT], coff[NIT], dst[NIT];     // row within the stage, column offset in floats (already + m0 / n0)
  bool on[NIT], isA[NIT], isA2[NIT];
  const int wa = min(T_BM, (A2 ? msplit + lda2 : lda) - m0), wb = min(BN, ldb - n0);      // columns that exist in memory
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int p = wave + G_WAVES * it;
    on[it] = p < NPIECE;
    const int pp = on[it] ? p : 0;
    isA[it] = pp < AP;
    const int q = isA[it] ? pp : pp - AP;
    const int pitch = isA[it] ? PA : PB;
    const int f = 256 * q + 4 * lane;                              // float index in the operand's stage image
    const int row = f / pitch, col = f % pitch;
    roff[it] = row;
    coff[it] = isA[it] ? m0 + (col < wa ? col : 0) : n0 + (col < wb ? col : 0);
    isA2[it] = isA[it] && A2 && coff[it] >= msplit;
    if (isA2[it]) coff[it] -= msplit;
    dst[it] = (isA[it] ? 0 : A_BYTES) + q * 1024;
  }
  auto dma_stage = [&](char* st, int k0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (on[it]) {
        const int k = min(k0 + roff[it], K - 1);                   // rows past the chunk are zeroed below
        const float* src = isA2[it] ? A2 + (size_t)k * lda2 + coff[it]
                                    : (isA[it] ? A + (size_t)k * lda + coff[it] : B + (size_t)k * ldb + coff[it]);
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(st + dst[it]), 16, 0, 0);
      }
  };

  const int li = lane & 15, kq = lane >> 4;
  const int a_off = (kq * PA + 80 * wm + li) * 4, b_off = A_BYTES + (kq * PB + WN * wn + li) * 4;
  auto compute = [&](const char* cur) {
#pragma unroll
    for (int kk = 0; kk < T_BK / 4; ++kk) {
      float a[5], b[T16];
#pragma unroll
      for (int i = 0; i < 5; ++i) a[i] = *(const float*)(cur + a_off + (4 * kk * PA + 16 * i) * 4);
#pragma unroll
      for (int j = 0; j < T16; ++j) b[j] = *(const float*)(cur + b_off + (4 * kk * PB + 16 * j) * 4);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < T16; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
    }
  };

  const int nk = (kend - kbeg + T_BK - 1) / T_BK;
  const int pitch = (No + 15) & ~15;                               // rows of the partials start 64-byte aligned (gemm32_tn_pitch)
  float* Pz = P + (size_t)z * Mo * pitch;
  auto enter = [&](int kt) -> char* {
    char* cur = smem + (kt & 1) * STAGE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // stage kt landed for all waves, stage kt-1 no longer being read
    const int live = kend - (kbeg + kt * T_BK);                    // rows of this stage inside the chunk
    if (live < T_BK) {                   // K tail (last stage of the last chunk only): zero A's dead rows
      for (int e = tid; e < (T_BK - live) * PA; e += 64 * G_WAVES) ((float*)cur)[live * PA + e] = 0.f;
      __syncthreads();
    }
    if (kt + 1 < nk) dma_stage(smem + ((kt + 1) & 1) * STAGE, kbeg + (kt + 1) * T_BK);
    return cur;
  };
  if (nk > 0) dma_stage(smem, kbeg);
#ifndef G32TN_EPI_FUSED
#define G32TN_EPI_FUSED 1
#endif
  // The chunk's LAST stage runs column by column of the wave's tile (all eight k groups of a column's five accumulators, then
  // the next column) and stores column j - 1 -- twenty 4-byte-per-lane store instructions -- between the MFMAs of column j, instead
  // of 20 T16 stores after the loop with the matrix pipe idle.  Per accumulator the MFMA order is unchanged: bit-identical.
  const int nloop = (G32TN_EPI_FUSED && nk > 0) ? nk - 1 : nk;
  for (int kt = 0; kt < nloop; ++kt) compute(enter(kt));
  if (G32TN_EPI_FUSED && nk > 0) {
    const char* cur = enter(nk - 1);
    float a[T_BK / 4][5];
#pragma unroll
    for (int kk = 0; kk < T_BK / 4; ++kk)
#pragma unroll
      for (int i = 0; i < 5; ++i) a[kk][i] = *(const float*)(cur + a_off + (4 * kk * PA + 16 * i) * 4);
    auto store1 = [&](int i, int j, int r) {
      const int col = n0 + WN * wn + 16 * j + li;
      const int row = m0 + 80 * wm + 16 * i + 4 * kq + r;
      if (col < No && row < Mo) Pz[(size_t)row * pitch + col] = acc[i][j][r];
    };
#pragma unroll
    for (int j = 0; j < T16; ++j) {
      float b[T_BK / 4];
#pragma unroll
      for (int kk = 0; kk < T_BK / 4; ++kk) b[kk] = *(const float*)(cur + b_off + (4 * kk * PB + 16 * j) * 4);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int kk = 0; kk < T_BK / 4; ++kk) {
          acc[i][j] = mfma16(a[kk][i], b[kk], acc[i][j]);
          if (j > 0 && (kk & 1)) store1(i, j - 1, kk >> 1);       // 4 stores of tile (i, j - 1) under the 8 MFMAs of tile (i, j)
        }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) store1(i, T16 - 1, r);
    return;
  }

#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < T16; ++j) {
      const int col = n0 + WN * wn + 16 * j + li;
      if (col >= No) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 80 * wm + 16 * i + 4 * kq + r;
        if (row < Mo) Pz[(size_t)row * pitch + col] = acc[i][j][r];
      }
    }
}

// out[Ro][Co] = zero-padded copy of W[R][C] (transpose: of W^T); bias, if given, goes to column C (the ones column's
// partner) of the non-transposed copy.
__global__ void pad_weight_kernel(const float* __restrict__ W, int R, int C, int transpose,
                                  const float* __restrict__ bias, float* __restrict__ out, int Ro, int Co) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)Ro * Co) return;
  const int r = (int)(idx / Co), c = (int)(idx % Co);
  float v = 0.f;
  if (transpose) {
    if (r < C && c < R) v = W[(size_t)c * C + r];
  } else if (r < R) {
    if (c < C) v = W[(size_t)r * C + c];
    else if (c == C && bias) v = bias[r];
  }
  out[idx] = v;
}

// big form: N slices of 64 T32 columns (T32 <= 7: 2 x 56 KB of B + 2 x 16 KB of A fill the CU's LDS);
// small form: N slices of 32 NW columns (NW <= 14 waves)
void nt_shape(int N, bool big, int& nsl, int& T) {
  const int t32 = cdiv_i(N, 32);
  nsl = cdiv_i(t32, 14);
  T = big ? cdiv_i(cdiv_i(t32, nsl), 2) : cdiv_i(t32, nsl);
}
// 128-row tiles once they give most CUs a workgroup of their own, 32-row tiles below that
bool nt_big(int M) { return cdiv_i(M, 128) >= 192; }

template <int T32>
int launch_ntp_t(const float* A, int lda, int M, int Kp, const float* Bp, float* C, int ldc, int N, hipStream_t st) {
  const int nm = cdiv_i(M, 128);
  const int nb_rows = gemm32_nt_rows(N);
  const size_t smem = 2 * (size_t)(128 + 64 * T32) * 128;
  static std::atomic<unsigned long long> done{0};
  if (ensure_dyn_smem((const void*)gemm32_ntp_kernel<T32>, smem, done) != WGNN_OK) return WGNN_ERR_HIP;
  static const std::string name = "gemm32_ntp_kernel<128x" + std::to_string(64 * T32) + ">";
  const double fl = 2.0 * M * (double)N * Kp;
  const double by = 4.0 * ((double)M * Kp + (double)N * Kp + (double)M * N);
  const int grid = nm < 256 ? nm : 256;                          // one workgroup per CU, its row tiles 256 apart
  PROF_LAUNCH(name.c_str(), fl, by, st,
              hipLaunchKernelGGL((gemm32_ntp_kernel<T32>), dim3(grid), dim3(64 * G_WAVES), smem, st, A, lda, M, Kp, Bp, nb_rows,
                                 C, ldc, N, nm));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

template <int MW, int NW, int T32>
int launch_nt_t(const float* A, int lda, int M, int Kp, const float* Bp, float* C, int ldc, int N, int nsl,
                hipStream_t st, int stagger = 0) {
  const int nm = cdiv_i(M, 32 * MW);
  const int nb_rows = gemm32_nt_rows(N);
  const size_t smem = 2 * (size_t)(32 * MW + 32 * T32 * NW) * 128;
  static std::atomic<unsigned long long> done{0};
  if (ensure_dyn_smem((const void*)gemm32_nt_kernel<MW, NW, T32>, smem, done) != WGNN_OK) return WGNN_ERR_HIP;
  static const std::string name = "gemm32_nt_kernel<" + std::to_string(32 * MW) + "x" + std::to_string(32 * T32 * NW) + ">";
  const double fl = 2.0 * M * (double)N * Kp;
  const double by = 4.0 * ((double)M * Kp + (double)N * Kp + (double)M * N);
  PROF_LAUNCH(name.c_str(), fl, by, st,
              hipLaunchKernelGGL((gemm32_nt_kernel<MW, NW, T32>), dim3(cdiv_i(nm, 8) * 8 * nsl), dim3(64 * MW * NW), smem,
                                 st, A, lda, M, Kp, Bp, nb_rows, C, ldc, N, nm, nsl, stagger));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

void tn_shape(int No, int& nNb, int& T) {
  const int t16 = cdiv_i(No, 16);
  nNb = cdiv_i(t16, 14);
  T = cdiv_i(cdiv_i(t16, nNb), 2);       // <= 7
}

template <int T16>
int launch_tn_t(const float* A, int lda, const float* B, int ldb, int K, int splitk, float* P, int Mo, int No, int nNb,
                const float* A2, int lda2, int msplit, hipStream_t st) {
  const int ntiles = cdiv_i(Mo, T_BM) * nNb;
  const int kchunk = cdiv_i(cdiv_i(K, splitk), T_BK) * T_BK;
  const size_t smem = 2 * (size_t)T_BK * (T_BM + 16 + 32 * T16 + 16) * 4;
  static std::atomic<unsigned long long> done{0};
  if (ensure_dyn_smem((const void*)gemm32_tn_kernel<T16>, smem, done) != WGNN_OK) return WGNN_ERR_HIP;
  static const std::string name = "gemm32_tn_kernel<" + std::to_string(T16) + ">";
  const double fl = 2.0 * Mo * (double)No * K;
  const double by = 4.0 * ((double)K * Mo + (double)K * No + (double)Mo * No * splitk);
  const int grid = cdiv_i(splitk, 8) * 8 * ntiles;
  PROF_LAUNCH(name.c_str(), fl, by, st,
              hipLaunchKernelGGL(gemm32_tn_kernel<T16>, dim3(grid), dim3(64 * G_WAVES), smem, st, A, lda, B, ldb, K,
                                 kchunk, splitk, P, Mo, No, nNb, ntiles, A2, lda2, msplit));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

}  // namespace

// Rows the padded B operand of an N-column product must have (whole workgroup tiles of either form).
int gemm32_nt_rows(int N) {
  int nsl, T, nsl2, T2;
  nt_shape(N, true, nsl, T);
  nt_shape(N, false, nsl2, T2);
  const int a = nsl * 64 * T, b = nsl2 * 32 * T2;
  return a > b ? a : b;
}

// The LDS-DMA kernels need the contraction padded to 32 and short enough for a single fp32 accumulation chain (the
// general kernel folds every 512).  Both pay off from a few thousand rows (NT: 32-row tiles; TN: K chunks of two stages).
bool gemm32_nt_supported(size_t BT, int Kp_f, int Kp_b) {
  return BT >= 4096 && Kp_f % 32 == 0 && Kp_b % 32 == 0 && Kp_f <= 512 && Kp_b <= 512;
}
bool gemm32_tn_supported(size_t BT) { return BT >= 4096; }

int launch_pad_weight(const float* W, int R, int C, int transpose, const float* bias, float* out, int Ro, int Co,
                      hipStream_t st) {
  const size_t n = (size_t)Ro * Co;
  PROF_LAUNCH("pad_weight_kernel", 0.0, 4.0 * (n + (double)R * C), st,
              hipLaunchKernelGGL(pad_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, R, C,
                                 transpose, bias, out, Ro, Co));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

// C[M][N] = A[M][Kp] Bp[.][Kp]^T; A rows 16-byte aligned (lda % 4 == 0), Bp with gemm32_nt_rows(N) zero-padded rows.
int launch_gemm32_nt(const float* A, int lda, int M, int Kp, const float* Bp, float* C, int ldc, int N, hipStream_t st) {
  const bool big = nt_big(M);
  int nsl, T;
  nt_shape(N, big, nsl, T);
  if (Kp % 32 != 0 || lda % 4 != 0 || ((uintptr_t)A & 15) || ((uintptr_t)Bp & 15)) return WGNN_ERR_SHAPE;
  const int form = opt_gemm32_form();
  if (big && form == 34 && nsl == 1) {       // persistent form (one slice of N)
    switch (T) {
#define NT_CASE(t) \
  case t: return launch_ntp_t<t>(A, lda, M, Kp, Bp, C, ldc, N, st);
      NT_CASE(1) NT_CASE(2) NT_CASE(3) NT_CASE(4) NT_CASE(5) NT_CASE(6) NT_CASE(7)
#undef NT_CASE
    }
    return WGNN_ERR_SHAPE;
  }
  if (big && form > 0 && form < 34) {
    // two 4-wave workgroups per CU: 128 x 32 T tiles with T <= 5 (2 x 72 KB of LDS), the N range cut into equal slices of whole
    // 32-column tiles (the last one may hold fewer)
    const int t32 = cdiv_i(N, 32), ns2 = cdiv_i(t32, 5), T2 = cdiv_i(t32, ns2);
    if (ns2 * T2 - t32 <= 1) switch (T2) {     // the kernel's last slice may be one tile short, not more
#define NT_CASE(t) \
  case t: return launch_nt_t<4, 1, t>(A, lda, M, Kp, Bp, C, ldc, N, ns2, st, form - 1);
      NT_CASE(1) NT_CASE(2) NT_CASE(3) NT_CASE(4) NT_CASE(5)
#undef NT_CASE
    }
  }
  if (big) {
    switch (T) {
#define NT_CASE(t) \
  case t: return launch_nt_t<4, 2, t>(A, lda, M, Kp, Bp, C, ldc, N, nsl, st);
      NT_CASE(1) NT_CASE(2) NT_CASE(3) NT_CASE(4) NT_CASE(5) NT_CASE(6) NT_CASE(7)
#undef NT_CASE
    }
    return WGNN_ERR_SHAPE;
  }
  switch (T) {
#define NT_CASE(t) \
  case t: return launch_nt_t<1, t, 1>(A, lda, M, Kp, Bp, C, ldc, N, nsl, st);
    NT_CASE(1) NT_CASE(2) NT_CASE(3) NT_CASE(4) NT_CASE(5) NT_CASE(6) NT_CASE(7) NT_CASE(8) NT_CASE(9) NT_CASE(10)
    NT_CASE(11) NT_CASE(12) NT_CASE(13) NT_CASE(14)
#undef NT_CASE
  }
  return WGNN_ERR_SHAPE;
}

// Workgroup tiles of the split-K product (for the split-K choice) and the product itself:
// P[z][Mo][gemm32_tn_pitch(No)] = sum over K chunk z of A[k][m] B[k][n]; rows of A and B 16-byte aligned; reduced by finish.hip (plain [z][Mo][No] layout).
int gemm32_tn_pitch(int No) { return (No + 15) & ~15; }           // row pitch of P[z][Mo][pitch]
int gemm32_tn_tiles(int Mo, int No) {
  int nNb, T;
  tn_shape(No, nNb, T);
  return cdiv_i(Mo, T_BM) * nNb;
}
int launch_gemm32_tn(const float* A, int lda, const float* B, int ldb, int K, int splitk, float* P, int Mo, int No,
                     const float* A2, int lda2, int msplit, hipStream_t st) {
  int nNb, T;
  tn_shape(No, nNb, T);
  if (lda % 4 != 0 || ldb % 4 != 0 || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || splitk < 1) return WGNN_ERR_SHAPE;
  if (A2 && (lda2 % 4 != 0 || msplit % 4 != 0 || msplit > lda || ((uintptr_t)A2 & 15))) return WGNN_ERR_SHAPE;
  switch (T) {
#define TN_CASE(t) \
  case t: return launch_tn_t<t>(A, lda, B, ldb, K, splitk, P, Mo, No, nNb, A2, lda2, msplit, st);
    TN_CASE(1) TN_CASE(2) TN_CASE(3) TN_CASE(4) TN_CASE(5) TN_CASE(6) TN_CASE(7)
#undef TN_CASE
  }
  return WGNN_ERR_SHAPE;
}
