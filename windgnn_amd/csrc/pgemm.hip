// Plane GEMMs: split-fp16 ("f16x3") MFMA GEMMs whose operands already live in HBM as fp16 hi/lo
// planes written by their producers (gcnx_fwd -> g, grux_bwd -> dGI/dGH, grux_fwd -> Y planes,
// split_weight -> W_ih).  No conversion work is left in the GEMMs: staging is a pure 16-byte copy.
//
//   NT  C[M][N] (fp32) = A[M][Kp] . B[N][Kp]^T          GI = g W_ih^T (+b_ih via the ones column),
//                                                       dg = dGI W_ih
//   TN  P[z][Mo][No]   = sum_k A[k][m] * B[k][n]        dW_ih = dGI^T [g|1], dW_hh = dGH^T [Hprev|1]
//
// Both kernels stage their operands with LDS-DMA (global_load_lds_dwordx4, swizzle applied on the
// source address) into a two-stage ring and run ONE large workgroup per CU: what bounds these kernels
// is the CU's L2 -> LDS path (~17-23 B/clk measured; SQ counters: matrix pipe 33-39 % busy, waves half
// their time waiting for the next stage), so tiles are as large as LDS and the register file allow to
// minimise the bytes staged per MFMA.  All MFMAs are v_mfma_f32_16x16x32_f16 (a 32x32x16 rebuild of the
// NT kernel measured 8 % slower: DESIGN.md section 5).
#include <string>

#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------------------
// NT.  One 12-wave workgroup per CU owns a 192 x 32T tile: waves 6 (M) x 2 (N), wave tile 32 x 16T
// built from v_mfma_f32_16x16x32_f16 (K step = one 64-byte LDS row).  The CU's L2 -> LDS load path
// is what bounds this kernel:
// the tile is as tall and as wide as LDS allows so that the fewest bytes cross L2 -> LDS per MFMA
// (A is staged once per M tile when N <= 320, B once per 192 rows).
constexpr int NT_BM = 192, NT_WAVES = 12;

// 16-byte chunk c of 64-byte row `row` sits at chunk c ^ f(row): conflict-free for the 16-lane
// groups of ds_read_b128 when lane l reads chunk l>>4 of row l&15.
__device__ __forceinline__ int swz16(int row) { return (4 - (row >> 2)) & 3; }
__device__ __forceinline__ int sw16_off(int row, int c) { return row * 64 + ((c ^ swz16(row)) << 4); }

__device__ __forceinline__ f32x4 mfma_q(h8 a, h8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// Slots of the A ring and the kernel's dynamic LDS.  A third A slot (an A stage then has two compute phases to arrive) pays
// only in the one-pass instances, whose compute phase is a third as long: pgemm_nt<10,f16> 46.0 -> 42.7 us, <14,f16> 52.2 ->
// 50.0; the split instances fit it too (T <= 10 with a lo plane of A) and measured 88.2 -> 90.3 / 67.1 -> 65.4 us (<10> / <14,x2>):
// not what they wait for, so they keep two
constexpr int nt_a_slot(bool x3, bool alo) { return ((x3 && alo) ? 2 : 1) * NT_BM * 64; }
constexpr int nt_b_slot(int T, bool x3) { return (x3 ? 2 : 1) * 32 * T * 64; }
constexpr int nt_a_depth(int T, bool x3, bool alo) {
  return (!x3 && 3 * nt_a_slot(x3, alo) + 2 * nt_b_slot(T, x3) <= 156 * 1024) ? 3 : 2;
}
constexpr size_t nt_smem(int T, bool x3, bool alo) {
  return (size_t)nt_a_depth(T, x3, alo) * nt_a_slot(x3, alo) + 2 * (size_t)nt_b_slot(T, x3);
}

// ALO (with X3): the A operand has a lo plane (three passes lo*hi + hi*lo + hi*hi); false: A is a single fp16 plane and
// the product is hi*lo + hi*hi (two passes, no A-lo staging) -- the backward's dGI at large B*T, see DESIGN.md section 3.
// OUT16: C is a single fp16 plane (row pitch ldc in halfs): an accumulator quad's value is exchanged with the neighbouring
// lane (DPP quad_perm [1,0,3,2]) so that every lane stores two packed (column, column + 1) pairs -- 32-byte segments per 16
// lanes, half the bytes of the fp32 epilogue.  Used for dg in WGNN_MATH_F16X3G (its consumer, the GCN backward, rounds dg to
// fp16 planes anyway; the single plane's rounding averages out in the conv gradients' sums over B*T*S rows).
template <int T, bool X3, bool ALO, bool OUT16>
__global__ void __launch_bounds__(64 * NT_WAVES) pgemm_nt_kernel(const _Float16* __restrict__ Ahi,
                                                                const _Float16* __restrict__ Alo, int lda, int M,
                                                                int Kp, const _Float16* __restrict__ Bpl, int Np,
                                                                float* __restrict__ C, int ldc, int N,
                                                                const float* __restrict__ s_out_p, int nm, int nsl,
                                                                size_t bplane, int kc_len, int kc_first,
                                                                size_t cstride, int accumulate) {
  // K chunking (long contractions): this block multiplies columns [k0, k0 + Kp) of the operands, chunk index
  // kc_first + blockIdx.y, into C + blockIdx.y * cstride (split-K partials) or straight into / onto C.
  {
    const int k0 = (kc_first + (int)blockIdx.y) * kc_len;
    Ahi += k0;
    Alo += k0;
    Bpl += (size_t)(k0 / 32) * Np * 32;
    C += (size_t)blockIdx.y * cstride;
    Kp = min(kc_len, Kp - k0);
  }
  constexpr int BM = NT_BM, BNW = 16 * T, BN = 2 * BNW;
  constexpr int A_PL = BM * 64, B_PL = BN * 64;
  constexpr int PL = X3 ? 2 : 1;                                   // B planes moved: hi (+ lo)
  constexpr int PLA = (X3 && ALO) ? 2 : 1;                         // A planes moved
  constexpr int AP = BM / 16, BP = BN / 16;                        // 1 KB pieces per plane
  static_assert(AP == NT_WAVES, "every wave moves exactly PLA pieces of A per stage (the vmcnt below counts them)");
  constexpr int NITB = (PL * BP + NT_WAVES - 1) / NT_WAVES;        // B pieces per wave and stage (the last one may be off)
  // Two rings: B (weights: L2-resident) two slots deep, A (the activation planes: HBM) DA slots deep -- three where LDS
  // allows, so an A stage has two compute phases to arrive instead of one
  constexpr int A_SLOT = PLA * A_PL, B_SLOT = PL * B_PL, DA = nt_a_depth(T, X3, ALO);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: LDS-DMA bases go to M0
  const int wm = wave % 6, wn = wave / 6;
  // blocks {b, b+8, ...} (same XCD, dispatched together) are the N slices of one M tile: the A tile is
  // fetched from HBM once and re-read from that XCD's L2 by the sibling slices.
  // (With fewer than 8 M tiles that padding would leave whole XCDs without work: plain order then.)
  int sl, mt;
  if (nm >= 8) {
    const int grp = blockIdx.x / (8 * nsl), within = blockIdx.x % (8 * nsl);
    sl = within / 8;
    mt = grp * 8 + within % 8;
  } else {
    sl = blockIdx.x / nm;
    mt = blockIdx.x % nm;
  }
  if (mt >= nm) return;
  const int m0 = mt * BM, n0 = sl * BN;
  const float s_out = s_out_p ? s_out_p[1] : 1.f;

  f32x4 acc[2][T];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- LDS-DMA staging (global_load_lds_dwordx4): no VGPRs, no ds_write.  One wave-instruction moves
  // a 1 KB "piece" = 16 tile rows x 64 B; the LDS image is lane-linear, so the swizzle the fragment
  // reads expect is applied to the SOURCE chunk index: LDS (row, pos) <- global chunk pos ^ f(row).
  const int prow = lane >> 2, ppos = lane & 3;
  const int chunk = 8 * (ppos ^ swz16(prow));
  const _Float16* srcA[PLA];                                         // wave w moves rows 16 w .. 16 w + 15 of every A plane
  const _Float16* srcB[NITB];
  int dstB[NITB];
  bool onB[NITB];
  {
    const int gr = min(m0 + 16 * wave + prow, M - 1);               // rows past M are computed but never stored
    srcA[0] = Ahi + (size_t)gr * lda + chunk;
    if (PLA == 2) srcA[PLA - 1] = Alo + (size_t)gr * lda + chunk;
  }
#pragma unroll
  for (int it = 0; it < NITB; ++it) {
    const int q = wave + NT_WAVES * it;
    onB[it] = q < PL * BP;                                          // wave-uniform
    const int plane = onB[it] ? q / BP : 0, blk = onB[it] ? q % BP : 0;
    srcB[it] = Bpl + (size_t)plane * bplane + (size_t)(n0 / 16 + blk) * 512 + lane * 8;   // one fragment: 1 KB, linear (bimg_off)
    dstB[it] = plane * B_PL + blk * 1024;
  }
  const int nk = Kp / 32;
  const int r16 = lane & 15, c4 = lane >> 4;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  char* const ringA = smem;
  char* const ringB = smem + DA * A_SLOT;

  auto dma_A = [&](int slot, int kt) {
#pragma unroll
    for (int pl = 0; pl < PLA; ++pl)
      __builtin_amdgcn_global_load_lds((glb_void*)(srcA[pl] + (size_t)32 * kt),
                                       (lds_void*)(ringA + slot * A_SLOT + pl * A_PL + wave * 1024), 16, 0, 0);
  };
  auto dma_B = [&](int slot, int kt) {
#pragma unroll
    for (int it = 0; it < NITB; ++it)
      if (onB[it])
        __builtin_amdgcn_global_load_lds((glb_void*)(srcB[it] + (size_t)Np * 32 * kt),
                                         (lds_void*)(ringB + slot * B_SLOT + dstB[it]), 16, 0, 0);
  };
  const int a_off0 = sw16_off(32 * wm + r16, c4), a_off1 = a_off0 + 16 * 64;
  const int b_off = T * wn * 1024 + lane * 16;      // B pieces are fragment-major: lane l owns bytes [16 l, 16 l + 16) of each
  auto compute = [&](const char* curA, const char* curB) {
    const char* Ah = curA;
    const char* Al = curA + A_PL;
    const char* Bh = curB + b_off;
    const char* Bl = Bh + B_PL;
    h8 ah[2], al[2];
    ah[0] = *(const h8*)(Ah + a_off0);
    ah[1] = *(const h8*)(Ah + a_off1);
    al[0] = ah[0];
    al[1] = ah[1];
    if (X3 && ALO) {
      al[0] = *(const h8*)(Al + a_off0);
      al[1] = *(const h8*)(Al + a_off1);
    }
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const h8 bh = *(const h8*)(Bh + j * 1024);
      if (X3) {
        const h8 bl = *(const h8*)(Bl + j * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (ALO) acc[i][j] = mfma_q(al[i], bh, acc[i][j]);
          acc[i][j] = mfma_q(ah[i], bl, acc[i][j]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = mfma_q(ah[i], bh, acc[i][j]);
    }
  };

  if constexpr (DA == 2) {
    // Two-stage ring: stage kt+1 is in flight while stage kt is multiplied.
    dma_A(0, 0);
    dma_B(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // all waves: stage kt visible, stage kt-1 no longer being read
      if (kt + 1 < nk) {
        dma_A((kt + 1) & 1, kt + 1);
        dma_B((kt + 1) & 1, kt + 1);
      }
      compute(ringA + (kt & 1) * A_SLOT, ringB + (kt & 1) * B_SLOT);
    }
  } else {
    // A three slots deep, B two.  Iteration kt issues B(kt+1), THEN A(kt+2): a wave's loads complete in issue order, so
    // "at most the PLA loads of A(kt+1) still outstanding" (vmcnt(PLA)) at the top of iteration kt means B(kt) and A(kt)
    // have landed while A(kt+1) is still on its way.  When there is no A(kt+1) (last stage) the wait is for everything.
    dma_A(0, 0);
    dma_B(0, 0);
    if (nk > 1) dma_A(1, 1);
    int sa = 0;                          // kt % 3
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {
        if (PLA == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();      // all waves: stage kt visible; nobody still reads stage kt-1 (its slots are reused now)
      if (kt + 1 < nk) dma_B((kt + 1) & 1, kt + 1);
      const int sa2 = sa == 0 ? 2 : sa - 1;             // (kt + 2) % 3
      if (kt + 2 < nk) dma_A(sa2, kt + 2);
      compute(ringA + sa * A_SLOT, ringB + (kt & 1) * B_SLOT);
      sa = sa == 2 ? 0 : sa + 1;
    }
  }

  // ---- epilogue through LDS (the ring is idle now): the accumulators of ROWS tile rows at a time are laid out row-major in
  // LDS and leave as 16-byte stores, 1 KB per wave-instruction -- a quarter of the store instructions of the per-lane form
  // below (dword stores in 64-byte segments), which is store-ISSUE-bound: measured 19-22 us of pgemm_nt<10>'s 87 with the
  // main loop ablated.  Taken when C rows are 16-byte aligned and nothing is accumulated onto C.
  {
    constexpr int PW = BN + 4;                                      // LDS row pitch in floats: rows 4 apart are 16 banks apart
    constexpr int ROWS = nt_smem(T, X3, ALO) >= (size_t)96 * PW * 4 ? 96 : 48;
    constexpr int NPASS = BM / ROWS;                                // 2 (pass = i) or 4 (pass = 2 i + (wm >= 3))
    static_assert(nt_smem(T, X3, ALO) >= (size_t)ROWS * PW * 4, "the ring holds one pass of the tile");
    constexpr int CW = OUT16 ? 8 : 4;                               // columns per 16-byte store
    const int ldb16 = OUT16 ? ldc / 8 * 8 : ldc / 4 * 4;
    // (fp32 C of the split instances: pgemm_nt<10> 89.3 -> 85.1 / 87.5 -> 85.2 / 87.2 -> 84.9 us; the packed fp16 epilogue and
    // the one-pass instances measured 0 ... +1 us with it and keep the per-lane form)
    if (X3 && !OUT16 && !accumulate && ldb16 == ldc && ((size_t)C & 15) == 0) {     // wave-uniform
      float* tile = (float*)smem;
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int i = NPASS == 2 ? pass : pass >> 1;
        const bool mine = NPASS == 2 || (wm >= 3) == (bool)(pass & 1);
        const int wl = NPASS == 2 ? wm : wm % 3;
        __syncthreads();                                            // the ring / the previous pass's rows are no longer read
        if (mine) {
#pragma unroll
          for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              tile[(16 * wl + 4 * c4 + r) * PW + BNW * wn + 16 * j + r16] = acc[i][j][r] * s_out;
        }
        __syncthreads();
        for (int q = tid; q < ROWS * (BN / CW); q += 64 * NT_WAVES) {
          const int lr = q / (BN / CW), cc = q % (BN / CW);
          const int wq = lr >> 4, x = lr & 15;
          const int row = m0 + 32 * (NPASS == 2 ? wq : wq + 3 * (pass & 1)) + 16 * i + x;
          const int col = n0 + CW * cc;
          if (row >= M || col >= N) continue;
          const float* src = tile + lr * PW + CW * cc;
          if (OUT16) {
            const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
            typedef _Float16 h8v __attribute__((ext_vector_type(8)));
            h8v h;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h[e] = (_Float16)a[e]; h[4 + e] = (_Float16)b[e]; }
            _Float16* dst = (_Float16*)C + (size_t)row * ldc + col;
            if (col + CW <= ldc) *(h8v*)dst = h;
            else for (int e = 0; e < CW && col + e < N; ++e) dst[e] = h[e];
          } else {
            const f32x4 a = *(const f32x4*)src;
            float* dst = C + (size_t)row * ldc + col;
            if (col + CW <= ldc) *(f32x4*)dst = a;
            else for (int e = 0; e < CW && col + e < N; ++e) dst[e] = a[e];
          }
        }
      }
      return;
    }
  }
  if (OUT16) {
    _Float16* Ch = (_Float16*)C;
    const bool odd = lane & 1;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const int col = n0 + BNW * wn + 16 * j + r16, colp = col & ~1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 v = acc[i][j] * s_out;
        // even lanes keep rows 0, 1 of the quad and take the partner's values of the same rows (column + 1);
        // odd lanes keep rows 2, 3 and take the partner's (column - 1)
        const float s0 = odd ? v[0] : v[2], s1 = odd ? v[1] : v[3];            // what the partner needs from me
        const float k0 = odd ? v[2] : v[0], k1 = odd ? v[3] : v[1];            // what I keep
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, false));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        const f2v a0 = {odd ? r0 : k0, odd ? k0 : r0}, a1 = {odd ? r1 : k1, odd ? k1 : r1};
        const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(a0, h2v));
        const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(a1, h2v));
        const int row = m0 + 32 * wm + 16 * i + 4 * c4 + (odd ? 2 : 0);
        if (colp < N) {
          if (row < M) *(unsigned*)(Ch + (size_t)row * ldc + colp) = p0;
          if (row + 1 < M) *(unsigned*)(Ch + (size_t)(row + 1) * ldc + colp) = p1;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const int col = n0 + BNW * wn + 16 * j + r16;
    if (col >= N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 32 * wm + 16 * i + 4 * c4 + r;
        if (row < M) {
          float* dst = C + (size_t)row * ldc + col;
          const float v = acc[i][j][r] * s_out;
          *dst = accumulate ? *dst + v : v;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// TN.  One 8-wave workgroup per CU owns a 320 x 32T tile of P for one K chunk: waves 4 (M) x 2 (N),
// wave tile 80 x 16T of v_mfma_f32_16x16x32_f16.  Both operands are K-strided in memory, so a stage
// (32 k rows) is staged row-major by LDS-DMA exactly as it lies in HBM and the fragments are formed
// by ds_read_b64_tr_b16 (hardware transpose read).  The MFMA's k slots of lane group x are rows
// {4x..4x+3} and {16+4x..16+4x+3} of the stage (any bijection works as long as A and B agree), so
// one transpose read of a half-wave touches 8 consecutive rows x 32 B; the 32-byte segments of a row
// are XOR-swizzled by g(row) (applied on the DMA source address) so that those 8 pieces fall into the
// 8 different 32-byte bank groups whatever the row length.
// TN_BM = 320, TN_WAVES = 8: common.h (finish.hip reads this kernel's partial layout)

__device__ __forceinline__ int tn_g(int stride32, int r) {   // stride32 = row bytes / 32 (even)
  const int u = stride32 & 7;
  return u == 0 ? (r & 7) : (u == 4 ? ((r >> 1) & 3) : ((r >> 2) & 1));
}

// A2 (template flag): the A operand's columns m >= msplit come from a second pair of planes, column m - msplit of
// A2hi / A2lo (row stride lda2); msplit is a multiple of 8 so no 16-byte chunk straddles the two sources.  Used for
// dW_hh = [dGI_r | dGI_z | dGH_n]^T Hprev: the BPTT kernel stores the n third of dGH only (its r and z thirds equal dGI's).
// ALO (with X3): as in pgemm_nt_kernel -- false: the A operand (dGI [+ dGHn]) is a single fp16 plane, two passes.
template <int T, bool X3, bool A2, bool ALO>
__global__ void __launch_bounds__(64 * TN_WAVES) pgemm_tn_kernel(const _Float16* __restrict__ Ahi,
                                                                const _Float16* __restrict__ Alo, int lda,
                                                                const _Float16* __restrict__ Bhi,
                                                                const _Float16* __restrict__ Blo, int ldb,
                                                                int shift_T, int K, int kchunk,
                                                                float* __restrict__ partial, int Mout, int Nout,
                                                                int nNb, const _Float16* __restrict__ A2hi,
                                                                const _Float16* __restrict__ A2lo, int lda2,
                                                                int msplit, int b_stream) {
  constexpr int BM = TN_BM, BN = 32 * T;
  constexpr int ARB = BM * 2, BRB = BN * 2;                         // row bytes
  constexpr int A_PL = 32 * ARB, B_PL = 32 * BRB, STAGE = 2 * A_PL + 2 * B_PL;
  constexpr int PL = X3 ? 2 : 1;                                   // B planes moved: hi (+ lo)
  constexpr int PLA = (X3 && ALO) ? 2 : 1;                         // A planes moved
  constexpr int AP = A_PL / 1024, BP = B_PL / 1024;                // 1 KB pieces per plane
  constexpr int NPIECE = PLA * AP + PL * BP, NIT = (NPIECE + TN_WAVES - 1) / TN_WAVES;
  constexpr int ACH = ARB / 16, BCH = BRB / 16;                    // 16-byte chunks per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int z = blockIdx.x;
  const int mb = blockIdx.y / nNb, nb = blockIdx.y % nNb;
  const int m0 = mb * BM, n0 = nb * BN;
  const int kbeg = z * kchunk, kend = min(K, kbeg + kchunk);
  const int nk = kend > kbeg ? (kend - kbeg + 31) / 32 : 0;

  f32x4 acc[5][T];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging map: piece p = wave + 8*it; lane-linear chunk q of a plane's stage image is (row q / CH,
  // position q % CH) and receives source chunk position ^ 2 g(row)
  const _Float16* base[NIT];
  int ld[NIT], dst[NIT], prow[NIT], pcol[NIT];
  bool on[NIT], isb[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int p = wave + TN_WAVES * it;
    on[it] = p < NPIECE;                                            // wave-uniform
    const int pp = on[it] ? p : 0;
    isb[it] = pp >= PLA * AP;
    if (!isb[it]) {
      const int plane = pp / AP, q = 64 * (pp % AP) + lane;
      const int row = q / ACH, pos = q % ACH;
      const int col = m0 + 8 * (pos ^ (2 * tn_g(ARB / 32, row)));
      base[it] = plane ? Alo : Ahi;
      ld[it] = lda;
      prow[it] = row;
      pcol[it] = col < lda ? col : 0;                              // columns past the planes: any finite data
      if (A2 && col >= msplit) {
        base[it] = plane ? A2lo : A2hi;
        ld[it] = lda2;
        pcol[it] = col - msplit < lda2 ? col - msplit : 0;
      }
      dst[it] = plane * A_PL + (pp % AP) * 1024;
    } else {
      const int q2 = pp - PLA * AP;
      const int plane = q2 / BP, q = 64 * (q2 % BP) + lane;
      const int row = q / BCH, pos = q % BCH;
      const int col = n0 + 8 * (pos ^ (2 * tn_g(BRB / 32, row)));
      base[it] = plane ? Blo : Bhi;
      ld[it] = ldb;
      prow[it] = row;
      pcol[it] = col < ldb ? col : 0;
      dst[it] = 2 * A_PL + plane * B_PL + (q2 % BP) * 1024;
    }
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  auto dma_stage = [&](char* st, int k0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (on[it]) {
        int k = min(k0 + prow[it], kend - 1);                      // rows past the chunk: zeroed in LDS below
        if (isb[it] && shift_T > 0) k = (k % shift_T) != 0 ? k - 1 : K;   // row K = the stored t = 0 row
        // b_stream: the B operand is an old, read-once tensor (the g plane, the Hprev planes): non-temporal, so that it does
        // not push the A operand's planes (dGI: read again by the other GEMMs) out of the Infinity Cache
        if (isb[it] && b_stream)
          __builtin_amdgcn_global_load_lds((glb_void*)(base[it] + (size_t)k * ld[it] + pcol[it]),
                                           (lds_void*)(st + dst[it]), 16, 0, 2);
        else
          __builtin_amdgcn_global_load_lds((glb_void*)(base[it] + (size_t)k * ld[it] + pcol[it]),
                                           (lds_void*)(st + dst[it]), 16, 0, 0);
      }
  };

  // ---- transpose-read addressing: lane 16x + 4q + p supplies row 4x + q, bytes 8p..8p+7 of the tile's
  // 32-byte segment (second read: 16 rows further down)
  const int x = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int r1 = 4 * x + q4;
  const int gA = 2 * tn_g(ARB / 32, r1), gB = 2 * tn_g(BRB / 32, r1);
  int a_off[5], b_off[T];
#pragma unroll
  for (int i = 0; i < 5; ++i) a_off[i] = r1 * ARB + 16 * ((2 * (5 * wm + i) + (p4 >> 1)) ^ gA) + 8 * (p4 & 1);
#pragma unroll
  for (int j = 0; j < T; ++j)
    b_off[j] = 2 * A_PL + r1 * BRB + 16 * ((2 * (T * wn + j) + (p4 >> 1)) ^ gB) + 8 * (p4 & 1);
  typedef __attribute__((address_space(3))) fh4 lds_fh4;
  auto trread = [&](const char* p0, int rb) -> h8 {
    const fh4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4*)p0);
    const fh4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4*)(p0 + 16 * rb));
    const h4 w0 = __builtin_bit_cast(h4, v0), w1 = __builtin_bit_cast(h4, v1);
    h8 r;
    r[0] = w0[0]; r[1] = w0[1]; r[2] = w0[2]; r[3] = w0[3];
    r[4] = w1[0]; r[5] = w1[1]; r[6] = w1[2]; r[7] = w1[3];
    return r;
  };
  auto compute = [&](const char* cur) {
    h8 ah[5], al[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      ah[i] = trread(cur + a_off[i], ARB);
      al[i] = ah[i];
      if (X3 && ALO) al[i] = trread(cur + A_PL + a_off[i], ARB);
    }
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const h8 bh = trread(cur + b_off[j], BRB);
      if (X3) {
        const h8 bl = trread(cur + B_PL + b_off[j], BRB);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          if (ALO) acc[i][j] = mfma_q(al[i], bh, acc[i][j]);
          acc[i][j] = mfma_q(ah[i], bl, acc[i][j]);
        }
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) acc[i][j] = mfma_q(ah[i], bh, acc[i][j]);
    }
  };

  if (nk > 0) dma_stage(smem, kbeg);
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * STAGE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // all waves: stage kt visible, stage kt-1 no longer being read
    if (kt + 1 < nk) dma_stage(smem + ((kt + 1) & 1) * STAGE, kbeg + 32 * (kt + 1));
    const int rows = kend - (kbeg + 32 * kt);
    if (rows < 32) {                   // last, partial stage of the chunk: rows >= kend contribute nothing
      const u32x4 zero = {0u, 0u, 0u, 0u};
      for (int c = tid; c < (32 - rows) * ACH; c += 64 * TN_WAVES) {
        *(u32x4*)(cur + rows * ARB + 16 * c) = zero;
        if (X3 && ALO) *(u32x4*)(cur + A_PL + rows * ARB + 16 * c) = zero;
      }
      __syncthreads();
    }
    compute(cur);
  }

  // Partials are stored in the accumulators' own layout, [z][tile][wave][i][j][lane][4]: every store is a
  // full 1 KB wave-instruction (row-major P would be 64-byte segments, 4x the store instructions);
  // finish.hip (seg_tn) undoes the permutation while it sums over z.
  float* P = partial + (((size_t)z * gridDim.y + blockIdx.y) * TN_WAVES + wave) * (5 * T * 256) + lane * 4;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) {
      *(f32x4*)(P + (i * T + j) * 256) = acc[i][j];
    }
}

// Planes of O[Rp][Cp] (O[r][c] = transpose ? W[c][r] : W[r][c]; optional extra column `bias_col`
// holding bias[r]; zero elsewhere) from fp32 W[R][C], in the B-image layout bimg_off(r, c, Rp) (common.h): stage-major, and
// fragment-major inside a stage.  Thread i writes the i-th 16-byte piece of each plane -- 8 consecutive k of one image row --
// so a wavefront stores 1 KB of consecutive memory per plane and (not transposed) reads 16 W rows x 128 consecutive bytes.
// (One 2-byte store per thread, rounds 1-5a: 2.2 ms for each 36 864 x 12 288 image of BASELINE configs[4], 1.65 TB/s.)
__global__ void split_weight2_kernel(const float* __restrict__ W, int R, int C, int transpose,
                                     const float* __restrict__ bias, int bias_col, _Float16* hi, _Float16* lo,
                                     int Rp, int Cp, unsigned* status) {
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  const size_t pi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= (size_t)Rp * Cp / 8) return;
  const int tile = (int)(pi >> 6), within = (int)(pi & 63);       // 1 KB fragments: [K step][16-row tile][k chunk][row][8]
  const int kt = tile / (Rp >> 4), nt = tile % (Rp >> 4);
  const int r = 16 * nt + (within & 15), c0 = 32 * kt + 8 * (within >> 4);
  float v[8];
  if (transpose) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (c0 + k < R && r < C) ? W[(size_t)(c0 + k) * C + r] : 0.f;
  } else if (r < R && c0 + 7 < C && (((size_t)r * C + c0) & 3) == 0) {
    const f32x4 a = *(const f32x4*)(W + (size_t)r * C + c0), b = *(const f32x4*)(W + (size_t)r * C + c0 + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = c0 + k;
      v[k] = 0.f;
      if (r < R && c < C) v[k] = W[(size_t)r * C + c];
      else if (bias && r < R && c == bias_col) v[k] = bias[r];
    }
  }
  h8v vh, vl;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bad |= out_of_fp16_range(v[k]);
    vh[k] = (_Float16)v[k];
    vl[k] = (_Float16)(v[k] - (float)vh[k]);
  }
  *(h8v*)(hi + pi * 8) = vh;
  *(h8v*)(lo + pi * 8) = vl;
  report_status(status, bad, WGNN_STATUS_WEIGHT_RANGE);
}

}  // namespace

int launch_split_weight2(const float* W, int R, int C, int transpose, const float* bias, int bias_col, void* planes,
                         int Rp, int Cp, unsigned* status, hipStream_t st) {
  _Float16* hi = (_Float16*)planes;
  _Float16* lo = hi + (size_t)Rp * Cp;
  const size_t n8 = (size_t)Rp * Cp / 8;
  if ((Rp & 15) || (Cp & 31) || (((uintptr_t)planes | ((size_t)Rp * Cp * 2)) & 15)) return WGNN_ERR_SHAPE;
  PROF_LAUNCH("split_weight2_kernel", 0.0, 4.0 * R * C + 4.0 * (double)Rp * Cp, st,
              hipLaunchKernelGGL(split_weight2_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, W, R, C, transpose,
                                 bias, bias_col, hi, lo, Rp, Cp, status));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

// C = sum over chunks of the split-K partials [nchunks][n] (fixed order), float4 per thread
__global__ void nt_ksum_kernel(const float* __restrict__ part, int nchunks, size_t n4, float* __restrict__ C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = ((const f32x4*)part)[i];
  for (int c = 1; c < nchunks; ++c) s += ((const f32x4*)part)[(size_t)c * n4 + i];
  ((f32x4*)C)[i] = s;
}

// NT tiling of N: nsl slices of 32T columns (T <= 14: two stages of (192 + 448) 128-byte rows are exactly the
// CU's 160 KB of LDS), as few slices as possible and as narrow as they can be -- unless that leaves most CUs
// without a workgroup (few rows: the per-step GEMMs of a wide GRU, small batches), in which case N is cut finer.
// K chunk of long contractions: <= 128 chained MFMA steps per fp32 accumulator, then the chunk's sum is added onto C (two-level
// summation).  4096 since round 4 (2048 before): every chunk of the 4096-station projections re-reads and re-writes the whole
// C (453 / 654 MB per chunk for GI / dg at B*T = 3072), so half the chunks = 11.7 GB less traffic and half the launches:
// configs[4] 142.3 -> 132.4 ms per step, with Y / gradients at full width still inside the same bounds
// (test_4096_station_config_at_full_width_H12288_against_host_fp64; a single 53 248-long chain measured 4e-5 / 2.5e-4)
#ifndef WGNN_NT_KC
#define WGNN_NT_KC 4096
#endif
constexpr int NT_KC = WGNN_NT_KC;
static int nt_chunks(int Kp) { return Kp > NT_KC + NT_KC / 2 ? cdiv_i(Kp, NT_KC) : 1; }
static void nt_shape(int M, int N, int Kp, bool splitk, int& nsl, int& T) {
  nsl = cdiv_i(N, 448);
  T = cdiv_i(cdiv_i(N, nsl), 32);
  const int nm = cdiv_i(M, NT_BM);
  if (nm * nsl * (splitk ? nt_chunks(Kp) : 1) < 192) {
    int want = cdiv_i(256, nm);                       // slices that would give every CU a workgroup
    if (want > cdiv_i(N, 32)) want = cdiv_i(N, 32);
    T = cdiv_i(cdiv_i(N, want), 32);
    nsl = cdiv_i(N, 32 * T);
  }
}
// floats of split-K scratch that make a few-row, long-K product (the per-step GEMMs of a wide GRU) fill the chip
size_t pgemm_nt_kpart_floats(int M, int ldc, int Kp) { return nt_chunks(Kp) > 1 ? (size_t)nt_chunks(Kp) * M * ldc : 0; }
// rows of the B planes: any tiling nt_shape can pick stays inside them (rows >= N are zero)
int pgemm_nt_np(int N) { return cdiv_i(N, 32) * 32 + 448 - 32; }

template <int T>
static int launch_nt_t(const void* Ahi, const void* Alo, int lda, int M, int Kp, const void* Bplanes, int Np, float* C,
                       int ldc, int N, const float* s_out, bool x3, int nsl, float* kpart, bool out16, hipStream_t st) {
  const bool alo = Alo != nullptr;                 // x3 with a single-plane A operand: two passes
  const int nm = cdiv_i(M, NT_BM);
  const int grid = (nm >= 8 ? cdiv_i(nm, 8) * 8 : nm) * nsl;
  constexpr size_t smem3 = nt_smem(T, true, true), smem2 = nt_smem(T, true, false), smem1 = nt_smem(T, false, true);
  const size_t smem = x3 ? (alo ? smem3 : smem2) : smem1;
  static std::atomic<unsigned long long> done{0}, done16{0}, done2{0}, done2h{0}, done16h{0};
  if (ensure_dyn_smem((const void*)pgemm_nt_kernel<T, false, true, true>, smem1, done16h) != WGNN_OK) return WGNN_ERR_HIP;
  if (ensure_dyn_smem((const void*)pgemm_nt_kernel<T, true, true, false>, smem3, done) != WGNN_OK) return WGNN_ERR_HIP;
  if (ensure_dyn_smem((const void*)pgemm_nt_kernel<T, true, false, false>, smem2, done2) != WGNN_OK) return WGNN_ERR_HIP;
  if (ensure_dyn_smem((const void*)pgemm_nt_kernel<T, true, false, true>, smem2, done2h) != WGNN_OK) return WGNN_ERR_HIP;
  if (ensure_dyn_smem((const void*)pgemm_nt_kernel<T, false, true, false>, smem1, done16) != WGNN_OK) return WGNN_ERR_HIP;
  static const std::string name = "pgemm_nt_kernel<" + std::to_string(T) + ">", name16 = "pgemm_nt_kernel<" + std::to_string(T) + ",f16>",
                           name2 = "pgemm_nt_kernel<" + std::to_string(T) + ",x2>";
  // Long contractions (the 4096-station projections: K = 53 248) run as NT_KC-wide K chunks: an fp32 accumulator
  // chain of at most NT_KC / 32 MFMA steps per chunk keeps the summation error at fp32-GEMM level.  With split-K scratch
  // the chunks are blocks of ONE launch (partials summed in fixed order); without, one launch per chunk adds onto C.
  // (a one-pass fp16 product with fp16 C is in the 5e-2 tolerance class anyway: one chunk of any length)
  const int nchunks = (out16 && !x3) ? 1 : nt_chunks(Kp), kc_len = nchunks > 1 ? NT_KC : Kp;
  const size_t bplane = (size_t)Np * Kp;
  const bool split = kpart && nchunks > 1;
  const int nlaunch = split ? 1 : nchunks;
  if (out16 && ((x3 && alo) || nchunks > 1 || (ldc & 1))) return WGNN_ERR_UNSUPPORTED;   // fp16 C: the x2 / f16 instances, one K chunk
  for (int c = 0; c < nlaunch; ++c) {
    const dim3 g(grid, split ? nchunks : 1);
    float* out = split ? kpart : C;
    const size_t cstride = split ? (size_t)M * ldc : 0;
    const double kk = split ? Kp : (Kp - c * kc_len < kc_len ? Kp - c * kc_len : kc_len);
    const double fl = 2.0 * M * (double)N * kk;
    const double by = (x3 && alo ? 4.0 : 2.0) * (double)M * kk + (x3 ? 4.0 : 2.0) * (double)Np * kk +
                      (out16 ? 2.0 : 4.0) * (double)M * N * (split ? nchunks : (c > 0 ? 2 : 1));
    if (x3 && alo)
      PROF_LAUNCH(name.c_str(), fl, by, st,
                  hipLaunchKernelGGL((pgemm_nt_kernel<T, true, true, false>), g, dim3(64 * NT_WAVES), smem, st, (const _Float16*)Ahi,
                                     (const _Float16*)Alo, lda, M, Kp, (const _Float16*)Bplanes, Np, out, ldc, N, s_out,
                                     nm, nsl, bplane, kc_len, c, cstride, c > 0 ? 1 : 0));
    else if (x3 && out16)
      PROF_LAUNCH(name2.c_str(), fl, by, st,
                  hipLaunchKernelGGL((pgemm_nt_kernel<T, true, false, true>), g, dim3(64 * NT_WAVES), smem, st, (const _Float16*)Ahi,
                                     (const _Float16*)Ahi, lda, M, Kp, (const _Float16*)Bplanes, Np, out, ldc, N, s_out,
                                     nm, nsl, bplane, kc_len, c, cstride, 0));
    else if (x3)
      PROF_LAUNCH(name2.c_str(), fl, by, st,
                  hipLaunchKernelGGL((pgemm_nt_kernel<T, true, false, false>), g, dim3(64 * NT_WAVES), smem, st, (const _Float16*)Ahi,
                                     (const _Float16*)Ahi, lda, M, Kp, (const _Float16*)Bplanes, Np, out, ldc, N, s_out,
                                     nm, nsl, bplane, kc_len, c, cstride, c > 0 ? 1 : 0));
    else if (out16)
      PROF_LAUNCH(name16.c_str(), fl, by, st,
                  hipLaunchKernelGGL((pgemm_nt_kernel<T, false, true, true>), g, dim3(64 * NT_WAVES), smem, st, (const _Float16*)Ahi,
                                     (const _Float16*)Ahi, lda, M, Kp, (const _Float16*)Bplanes, Np, out, ldc, N, s_out,
                                     nm, nsl, bplane, kc_len, c, cstride, 0));
    else
      PROF_LAUNCH(name16.c_str(), fl, by, st,
                  hipLaunchKernelGGL((pgemm_nt_kernel<T, false, true, false>), g, dim3(64 * NT_WAVES), smem, st, (const _Float16*)Ahi,
                                     (const _Float16*)Alo, lda, M, Kp, (const _Float16*)Bplanes, Np, out, ldc, N, s_out,
                                     nm, nsl, bplane, kc_len, c, cstride, c > 0 ? 1 : 0));
    WGNN_CHECK_LAUNCH();
  }
  if (split) {
    if (((size_t)M * ldc) % 4 != 0) return WGNN_ERR_SHAPE;
    const size_t n4 = (size_t)M * ldc / 4;
    PROF_LAUNCH("nt_ksum_kernel", 0.0, 16.0 * n4 * (nchunks + 1), st,
                hipLaunchKernelGGL(nt_ksum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, kpart, nchunks, n4,
                                   C));
    WGNN_CHECK_LAUNCH();
  }
  return WGNN_OK;
}

// C[M][N] = s_out * A B^T.  A planes [M][lda] (Kp <= lda), B stage-major planes with Np = pgemm_nt_np(N) rows.
int launch_pgemm_nt(const void* Ahi, const void* Alo, int lda, int M, int Kp, const void* Bplanes, int Np, float* C,
                    int ldc, int N, const float* s_out, bool x3, float* kpart, hipStream_t st, bool out16, void* aimg) {
  // many rows AND many columns AND a long contraction (configs[4]'s projections): the 256 x 256 / 32x32x16 kernel of
  // pgemm_big.hip, one launch per K chunk (chunk sums added onto C, as below)
  if (x3 && !out16 && !kpart && !s_out && aimg && pgemm_nt256_wanted(M, N, Kp) && Kp % 32 == 0 && lda % 8 == 0 &&
      Np >= cdiv_i(N, 256) * 256 && opt_big_gemm()) {
    const int nchunks = nt_chunks(Kp), kc_len = nchunks > 1 ? NT_KC : Kp;
    // A as an image (every LDS-DMA piece of it 1 KB contiguous, half stages addressable): one pass over its planes
    int rc = launch_pgemm_repack_a(Ahi, Alo, lda, M, Kp, aimg, st);
    if (rc != WGNN_OK) return rc;
    const void* al = Alo ? (const void*)((const _Float16*)aimg + (size_t)cdiv_i(M, 256) * 256 * Kp) : nullptr;
    for (int k0 = 0; k0 < Kp;) {
      int klen = Kp - k0 < kc_len ? Kp - k0 : kc_len;
      if (Kp - (k0 + klen) < 256) klen = Kp - k0;          // a short tail (Kp = 53 280 = 13 x 4096 + 32) rides with the last full chunk
      rc = launch_pgemm_nt256(aimg, al, M, k0, klen, Bplanes, Np, (size_t)Np * Kp, C, ldc, N, k0 > 0, st);
      if (rc != WGNN_OK) return rc;
      k0 += klen;
    }
    (void)nchunks;
    return WGNN_OK;
  }
  int nsl, T;
  nt_shape(M, N, Kp, kpart != nullptr, nsl, T);
  if (Kp % 32 != 0 || lda % 8 != 0 || Np < nsl * 32 * T) return WGNN_ERR_SHAPE;
  switch (T) {
#define NT_CASE(t) \
  case t: return launch_nt_t<t>(Ahi, Alo, lda, M, Kp, Bplanes, Np, C, ldc, N, s_out, x3, nsl, kpart, out16, st);
    NT_CASE(1) NT_CASE(2) NT_CASE(3) NT_CASE(4) NT_CASE(5) NT_CASE(6) NT_CASE(7) NT_CASE(8) NT_CASE(9) NT_CASE(10)
    NT_CASE(11) NT_CASE(12) NT_CASE(13) NT_CASE(14)
#undef NT_CASE
  }
  return WGNN_ERR_SHAPE;
}

// TN tiling: M blocks of 320, N blocks of 32T columns (T <= 7), as few as possible and as narrow as they can be.
static void tn_shape(int Nout, int& nNb, int& T) {
  nNb = cdiv_i(Nout, 224);
  T = cdiv_i(cdiv_i(Nout, nNb), 32);
}
int pgemm_tn_tiles(int Mout, int Nout) {
  int nNb, T;
  tn_shape(Nout, nNb, T);
  return cdiv_i(Mout, TN_BM) * nNb;
}

template <int T>
static int launch_tn_t(const void* Ahi, const void* Alo, int lda, const void* Bhi, const void* Blo, int ldb, int shift_T,
                       int K, int splitk, float* partial, int Mout, int Nout, bool x3, int nNb, const void* A2hi,
                       const void* A2lo, int lda2, int msplit, hipStream_t st, bool b_stream) {
  const int nMb = cdiv_i(Mout, TN_BM);
  const int kchunk = cdiv_i(cdiv_i(K, splitk), 32) * 32;
  const size_t smem = 2 * (size_t)(2 * 32 * 2 * (TN_BM + 32 * T));
  const bool alo = Alo != nullptr;                 // x3 with a single-plane A operand (Alo == nullptr): two passes
  const double fl = 2.0 * Mout * (double)Nout * K;
  const double by = (x3 && alo ? 4.0 : 2.0) * (double)K * Mout + (x3 ? 4.0 : 2.0) * (double)K * Nout +
                    4.0 * (double)splitk * Mout * Nout;
  static const std::string name = "pgemm_tn_kernel<" + std::to_string(T) + ">",
                           name16 = "pgemm_tn_kernel<" + std::to_string(T) + ",f16>",
                           name2 = "pgemm_tn_kernel<" + std::to_string(T) + ",x2>";
  const dim3 grid(splitk, nMb * nNb), block(64 * TN_WAVES);
  if (!alo) { Alo = Ahi; A2lo = A2hi; }            // never read
#define TN_ARGS (const _Float16*)Ahi, (const _Float16*)Alo, lda, (const _Float16*)Bhi, (const _Float16*)Blo, ldb, shift_T, K, \
                kchunk, partial, Mout, Nout, nNb, (const _Float16*)A2hi, (const _Float16*)A2lo, lda2, msplit, (int)b_stream
#define TN_GO(NAME, X3V, A2V, ALOV)                                                                                  \
  do {                                                                                                               \
    static std::atomic<unsigned long long> done_{0};                                                                 \
    if (ensure_dyn_smem((const void*)pgemm_tn_kernel<T, X3V, A2V, ALOV>, smem, done_) != WGNN_OK) return WGNN_ERR_HIP; \
    PROF_LAUNCH(NAME.c_str(), fl, by, st,                                                                            \
                hipLaunchKernelGGL((pgemm_tn_kernel<T, X3V, A2V, ALOV>), grid, block, smem, st, TN_ARGS));           \
  } while (0)
  if (A2hi) {
    if constexpr (T <= 4) {     // only the narrow dW_hh product has a two-source A operand
      if (x3 && alo) TN_GO(name, true, true, true);
      else if (x3) TN_GO(name2, true, true, false);
      else TN_GO(name16, false, true, true);
    } else {
      return WGNN_ERR_UNSUPPORTED;
    }
  } else if (x3 && alo) {
    TN_GO(name, true, false, true);
  } else if (x3) {
    TN_GO(name2, true, false, false);
  } else {
    TN_GO(name16, false, false, true);
  }
#undef TN_GO
#undef TN_ARGS
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

void pgemm_tn_geom(int Mgemm, int Nout, int* T, int* nNb, int* ntiles) {
  tn_shape(Nout, *nNb, *T);
  *ntiles = cdiv_i(Mgemm, TN_BM) * *nNb;
}

size_t pgemm_tn_partial_floats(int Mout, int Nout, int splitk) {
  int nNb, T;
  tn_shape(Nout, nNb, T);
  return (size_t)splitk * cdiv_i(Mout, TN_BM) * nNb * TN_WAVES * 5 * T * 256;
}

// partial (pgemm_tn_kernel's own layout, read by finish.hip; pgemm_tn_partial_floats floats) = per-K-chunk sums of A[k][m] B[k][n].
// A planes [K][lda], B planes [K][ldb].
// shift_T > 0: B row k is taken from row k-1, and from the extra row K (which the producer fills with what the
// operand looks like at a window start) where k % shift_T == 0.
int launch_pgemm_tn(const void* Ahi, const void* Alo, int lda, const void* Bhi, const void* Blo, int ldb, int shift_T,
                    int K, int splitk, float* partial, int Mout, int Nout, bool x3, const void* A2hi, const void* A2lo,
                    int lda2, int msplit, hipStream_t st, bool b_stream) {
  if (lda % 8 != 0 || ldb % 8 != 0 || K < 1) return WGNN_ERR_SHAPE;
  if (A2hi && (lda2 % 8 != 0 || msplit % 8 != 0 || msplit > lda)) return WGNN_ERR_SHAPE;
  int nNb, T;
  tn_shape(Nout, nNb, T);
  switch (T) {
#define TN_CASE(t) \
  case t: return launch_tn_t<t>(Ahi, Alo, lda, Bhi, Blo, ldb, shift_T, K, splitk, partial, Mout, Nout, x3, nNb, A2hi, A2lo, \
                                lda2, msplit, st, b_stream);
    TN_CASE(1) TN_CASE(2) TN_CASE(3) TN_CASE(4) TN_CASE(5) TN_CASE(6) TN_CASE(7)
#undef TN_CASE
  }
  return WGNN_ERR_SHAPE;
}
