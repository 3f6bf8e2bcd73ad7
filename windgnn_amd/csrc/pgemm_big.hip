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
//   * the A operand as an IMAGE too: a pass of its own (repack_a_kernel, 1.3 GB of traffic for g at
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

// Both operands are IMAGES (bimg_off): A's planes are rewritten by repack_a_kernel, B's are the prepared weight images.
// LDS is a ring of NS HALF-stage slots (a half stage = 16 of a K step's 32 k = one v_mfma_f32_32x32x16 deep: A 256 rows x 32 B
// per plane + B the same = 32 KB with two A planes): 5 slots = all 160 KB (6 of 24 KB for the single-plane A).  Why half
// stages: with the two-stage ring of 64 KB K steps (round 5's first two cuts, sequential and software-pipelined: 60.5 ms per
// step for GI + dg both, like the kernel they were to replace) a stage had ONE K step to arrive; SQ counters put the matrix
// pipe at 53 % busy and the waves at 49 % in s_waitcnt / barrier with an L2 hit rate of 71 % -- every K step ends up waiting
// for the slowest of its 64 pieces, i.e. for one trip beyond L2 (profiles/r5_c5_big_gemm.txt).  Half-stage slots keep NS - 2 = 3
// half steps of requests in flight behind the one being multiplied.
template <bool ALO>
__global__ void __launch_bounds__(64 * BG_WAVES) pgemm_nt256_kernel(const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo,
                                                                   int Mp, int M, int Kp, const _Float16* __restrict__ Bpl,
                                                                   int Np, size_t bplane, float* __restrict__ C, int ldc, int N,
                                                                   int nmt, int nnt, int accumulate) {
  constexpr int PLA = ALO ? 2 : 1;
  constexpr int HP = 256 * 32;                                    // bytes of one plane of a half stage: 256 rows x 16 halfs
  constexpr int A_SLOT = PLA * HP, SLOT = A_SLOT + 2 * HP;
  constexpr int NS = ALO ? 5 : 6;                                 // ring slots: 5 x 32 KB / 6 x 24 KB
  constexpr int P = PLA + 2;                                      // LDS-DMA pieces per wave and half stage
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

  // ---- LDS-DMA.  A 1 KB fragment of an image = 16 rows x 32 k as [k chunk][row][8 halfs]; half s of it (k chunks 2 s, 2 s + 1) is
  // 512 contiguous bytes.  One wave-instruction (1 KB) moves the halves of TWO neighbouring fragments: lanes 0-31 fragment f,
  // lanes 32-63 fragment f + 1.  Wave w moves piece w (rows 32 w .. 32 w + 31 of the tile) of every plane: P loads per half stage.
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  const int f2 = lane >> 5, l32 = lane & 31;
  const _Float16* srcA[PLA];
  const _Float16* srcB[2];
  {
    const size_t fa = (size_t)(m0 / 16 + 2 * wave + f2) * 512 + l32 * 8;
    srcA[0] = Ahi + fa;
    if (ALO) srcA[PLA - 1] = Alo + fa;
    const size_t fb = (size_t)(n0 / 16 + 2 * wave + f2) * 512 + l32 * 8;
    srcB[0] = Bpl + fb;
    srcB[1] = Bpl + bplane + fb;
  }
  const size_t astep = (size_t)Mp * 32, bstep = (size_t)Np * 32;   // halfs per K step of the images
#ifndef BG_ABLATE
#define BG_ABLATE 0      // tools/gemm256_ablate.hip builds this file with parts switched off (bit 0: no steady-state DMA, bit 1: no
#endif                   // fragment reads after the first, bit 2: no MFMAs, bit 3: no barriers); the library always builds 0
  auto dma = [&](int slot, int hs) {                               // half step hs = 2 kt + s
    if ((BG_ABLATE & 1) && hs >= NS) return;
    char* base = smem + slot * SLOT + wave * 1024;
    const size_t ao = (size_t)(hs >> 1) * astep + (hs & 1) * 256, bo = (size_t)(hs >> 1) * bstep + (hs & 1) * 256;
#pragma unroll
    for (int pl = 0; pl < PLA; ++pl)
      __builtin_amdgcn_global_load_lds((glb_void*)(srcA[pl] + ao), (lds_void*)(base + pl * HP), 16, 0, 0);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      __builtin_amdgcn_global_load_lds((glb_void*)(srcB[pl] + bo), (lds_void*)(base + A_SLOT + pl * HP), 16, 0, 0);
  };
  // fragment reads: a plane's half slot is [piece = row >> 5][fragment = (row >> 4) & 1][k chunk 0 / 1][row & 15][16 B]; lane l of the
  // 32x32x16 MFMA reads row l & 31, k chunk l >> 5 (ds_read_b128: conflict-free, both 16-lane groups of a half-wave read
  // 256 contiguous bytes)
  const int r32 = lane & 31, hsel = lane >> 5;
  const int laneoff = (r32 >> 4) * 512 + hsel * 256 + (r32 & 15) * 16;
  const int offA = 4 * wm * 1024 + laneoff, offB = A_SLOT + 2 * wn * 1024 + laneoff;
  struct Frags { h8 ah[4], al[4], bh[2], bl[2]; };
  bool first_load = true;
  auto load = [&](Frags& f, int slot) {
    if (BG_ABLATE & 2) {
      if (!first_load) return;
      first_load = false;
      f = Frags{};
    }
    const char* st = smem + slot * SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f.bh[j] = *(const h8*)(st + offB + j * 1024);
      f.bl[j] = *(const h8*)(st + offB + HP + j * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f.ah[i] = *(const h8*)(st + offA + i * 1024);
      if (ALO) f.al[i] = *(const h8*)(st + offA + HP + i * 1024);
    }
  };
  auto mm = [&](const Frags& f) {
    if (BG_ABLATE & 4) return;
    // pass-major: the three products of one accumulator are eight MFMAs apart (BG_ABLATE & 32: accumulator-major, dependent triples)
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (BG_ABLATE & 32) {
            if (pass != 0) continue;
            if (ALO) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
          } else if (pass == 0) {
            if (ALO) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
          } else if (pass == 1) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
          }
        }
  };
  // s_waitcnt vmcnt(n) lgkmcnt(0) through the builtin (the compiler's wait-count model sees it; an asm statement made it put
  // lgkmcnt(0) waits of its own in front of the MFMAs): simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14
#define BG_WAIT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | 0x70 | (((n) >> 4) << 14))

  // Half step h lives in slot h % NS.  Half step h, ONE scheduling region between two barriers:
  //   * the LDS-DMA of half step h - 1 + NS (into the slot half step h - 1 left: its fragments went to registers a step ago),
  //   * the fragment reads of half step h + 1 (its slot was declared landed by the barrier that ended half step h - 1),
  //   * the 24 MFMAs of half step h out of registers,
  // INTERLEAVED (sched_group_barrier: one DS read behind each of the first 12 MFMAs, one DMA behind every second of the next
  // 8): issued as a block in front of the MFMAs they cost the matrix pipe their issue time, because the two waves of a SIMD run
  // in step between barriers (ablation, tools/gemm256_ablate.hip: MFMAs alone 1.92 ms per K chunk, + fragment reads 2.40, + DMA
  // 2.67 -- the memory instructions' time ADDED to the MFMAs'; barriers removed: no change).  Then: fragments of h + 1 in
  // registers, half step h + 2 landed (its NS - 3 successors may stay in flight), barrier.  Every request has NS - 2 half steps
  // to arrive.  Past the end the DMA re-requests the last half step into a slot nobody reads again and the fragment read
  // re-reads a landed slot (no conditional memory instruction: a load on one of two merging paths makes the compiler's own
  // waits conservative).
  const int nh = Kp / 16;                                          // the launcher guarantees nh >= NS and nh even
#pragma unroll
  for (int q = 0; q < NS - 1; ++q) dma(q, q);
  BG_WAIT((NS - 2) * P);                                           // half step 0 has landed (in-order completion per wave)
  __builtin_amdgcn_s_barrier();
  Frags F0, F1;
  load(F0, 0);
  __builtin_amdgcn_sched_barrier(0);
  BG_WAIT((NS - 3) * P);                                           // half step 1 has landed, F0 is in registers
  __builtin_amdgcn_s_barrier();
  auto step = [&](const Frags& cur, Frags& nxt, int slot_frag, int slot_dma, int hs_dma) {
    dma(slot_dma, hs_dma < nh ? hs_dma : nh - 1);
    load(nxt, slot_frag);
    mm(cur);
    if (!(BG_ABLATE & 16)) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // 1 DS read
      }
#pragma unroll
      for (int k = 0; k < P; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);         // 2 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // 1 VMEM read (the LDS-DMA)
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);          // the rest
    }
    BG_WAIT((NS - 3) * P);
    if (!(BG_ABLATE & 8)) __builtin_amdgcn_s_barrier();
  };
  int s0 = 0;                                                      // h % NS, h even
  for (int h = 0; h < nh; h += 2) {
    const int sm1 = s0 == 0 ? NS - 1 : s0 - 1;                     // (h - 1) % NS
    const int s1 = s0 + 1 == NS ? 0 : s0 + 1, s2 = s1 + 1 == NS ? 0 : s1 + 1;
    step(F0, F1, s1, sm1, h - 1 + NS);
    step(F1, F0, h + 2 < nh ? s2 : s1, s0, h + NS);
    s0 = s2;
  }
  BG_WAIT(0);                                                      // (requests past the end)
#undef BG_WAIT

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

// One K chunk: C (+)= A[:, k0 : k0 + klen] . B[:, k0 : k0 + klen]^T.  Aimg_hi / Aimg_lo: the planes of the image
// launch_pgemm_repack_a wrote (lo may be NULL: single-plane A, two passes); B as for launch_pgemm_nt; Np >= the N tiles' rows
// (pgemm_nt_np(N) is).
int launch_pgemm_nt256(const void* Aimg_hi, const void* Aimg_lo, int M, int k0, int klen, const void* Bplanes, int Np,
                       size_t bplane, float* C, int ldc, int N, bool accumulate, hipStream_t st) {
  if (klen % 32 != 0 || k0 % 32 != 0 || klen < 128) return WGNN_ERR_SHAPE;
  const int nmt = cdiv_i(M, BG_BM), nnt = cdiv_i(N, BG_BN);
  if (Np < nnt * BG_BN) return WGNN_ERR_SHAPE;
  const bool alo = Aimg_lo != nullptr;
  const size_t smem = alo ? (size_t)5 * 4 * 8192 : (size_t)6 * 3 * 8192;
  const dim3 grid(8 * nmt * cdiv_i(nnt, 8)), block(64 * BG_WAVES);
  const int Mp = nmt * BG_BM;
  const size_t aoff = (size_t)(k0 / 32) * Mp * 32;
  const _Float16* ah = (const _Float16*)Aimg_hi + aoff;
  const _Float16* al = alo ? (const _Float16*)Aimg_lo + aoff : ah;
  const _Float16* bp = (const _Float16*)Bplanes + (size_t)(k0 / 32) * Np * 32;
  const double fl = 2.0 * M * (double)N * klen;
  const double by = (alo ? 4.0 : 2.0) * (double)M * klen + 4.0 * (double)N * klen + 4.0 * (double)M * N * (accumulate ? 2 : 1);
  static std::atomic<unsigned long long> done3{0}, done2{0};
  if (alo) {
    if (ensure_dyn_smem((const void*)pgemm_nt256_kernel<true>, smem, done3) != WGNN_OK) return WGNN_ERR_HIP;
    PROF_LAUNCH("pgemm_nt256_kernel", fl, by, st,
                hipLaunchKernelGGL((pgemm_nt256_kernel<true>), grid, block, smem, st, ah, al, Mp, M, klen, bp, Np, bplane, C, ldc, N,
                                   nmt, nnt, accumulate ? 1 : 0));
  } else {
    if (ensure_dyn_smem((const void*)pgemm_nt256_kernel<false>, smem, done2) != WGNN_OK) return WGNN_ERR_HIP;
    PROF_LAUNCH("pgemm_nt256_kernel<x2>", fl, by, st,
                hipLaunchKernelGGL((pgemm_nt256_kernel<false>), grid, block, smem, st, ah, al, Mp, M, klen, bp, Np, bplane, C, ldc, N,
                                   nmt, nnt, accumulate ? 1 : 0));
  }
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
