// Two-layer graph convolution, split-fp16 ("f16x3") MFMA, register-chained.
//
// Reference: the two GraphConvLayer calls of GCN_GRU.forward
// (src/step6_gcn_gru_combined_model.py:17-20; layer = src/step5_gcn_layer_model.py:13-23) and
// their autograd backward (src/main.py:79).
//
// One wavefront owns one (window, timestep) tile X_t [S,13] end to end and never touches LDS in the
// forward: every product is a v_mfma_f32_16x16x32_f16 whose accumulator tile (C layout: lane =
// column, 4 registers = 4 consecutive rows) is converted to fp16 hi/lo in registers and fed straight
// back as the A or B operand of the next product.  That works for any product that contracts over
// the ROW index of the accumulator stack; the constant operand (A, A^T, W) is pre-permuted once per
// wave to the k-order the accumulator registers impose:
//     k-slot (g = lane>>4, j) of K-step ks  <->  row rho = 16*(2*ks + (j>>2)) + 4*g + (j&3).
//
// Forward chain (all [rows][cols<=16] stacks):
//   U1[s'][f'] = X W1            X loaded straight from HBM as the A operand (natural k = f)
//   H1t[f][s]  = relu(U1^T A^T + b1)        (U1 as A operand, contraction over s')
//   U2[s][f']  = H1 W2                     (H1t as A operand, contraction over f)
//   g[s][f']   = relu(A U2 + b2), produced as g^T[f'][s] and stored station-major.
// Backward chain: see gcnx_bwd_kernel.
#include "common.h"
#include <type_traits>

#include "gcnx_dev.h"

namespace {

constexpr int FWD_WAVES = 8;   // waves per forward block (A fragments are shared through LDS)

// IO: X is 16-bit (fp16 / bf16 by `io`); false = fp32 (no conversion code at all in the default instance)
template <int NT, bool X3, bool IO>
__global__ void __launch_bounds__(64 * FWD_WAVES) gcnx_fwd_kernel(int ntiles, int S, const float* __restrict__ A,
                                                       const void* __restrict__ X, const void* __restrict__ xtail, int io,
                                                       const float* __restrict__ W1,
                                                       const float* __restrict__ b1, const float* __restrict__ W2,
                                                       const float* __restrict__ b2, _Float16* __restrict__ ghi,
                                                       _Float16* __restrict__ glo, int ldp, unsigned* status) {
  constexpr int KS = (NT + 1) / 2;
  constexpr int SP = 16 * NT;
  constexpr int NP = (SP * F13 / 2 + 63) / 64;
  // Range check (status block, include/windgnn.h).  A pre-activation z is inf / NaN exactly when an operand overflowed
  // fp16 upstream, and the ReLU's fmaxf(NaN, 0) = 0 would hide that: relu_nan() adds 0 * z, which is +-0 for finite z
  // and NaN otherwise, so the poison reaches the stored value; there chk collects 0 * (v - hi), NaN when v was NaN or
  // rounded to the fp16 infinity.  (Compares or a checksum over the pre-activations themselves cost a wave of
  // occupancy: the compiler sank them behind the tile's products and kept all 24 operands alive, 122 -> 160 VGPRs.)
  float chk = 0.f;
  static_assert(128 * NP >= (SP * F13 + 1 + 31) / 32 * 32, "pair map must cover the padded row");
  constexpr int NF = NT * KS;
  __shared__ __attribute__((aligned(16))) h8 sCA[2 * NF * 64];   // A fragments [frag][hi|lo][lane]
  __shared__ __attribute__((aligned(16))) float sbuf[FWD_WAVES * 2 * SP * XS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int I = S * F13;
  float* xb = sbuf + wave * 2 * SP * XS;
  float* ob = xb + SP * XS;                           // output staging [s][XS] fp32 (one b128 store per n-tile)
  for (int i = lane; i < 2 * SP * XS; i += 64) xb[i] = 0.f;   // pads (f >= 13, s >= S) stay zero forever
  if (wave == 0) {
    Frag T[NT][KS];
    build_A_frags<NT, KS, false, X3>(T, A, S, c, g);
#pragma unroll
    for (int mi = 0; mi < NT; ++mi)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sCA[((mi * KS + ks) * 2 + 0) * 64 + lane] = T[mi][ks].hi;
        sCA[((mi * KS + ks) * 2 + 1) * 64 + lane] = T[mi][ks].lo;
      }
  }
  __syncthreads();
  // The half fragments (odd NT: the last k step covers 16 stations, 4 halfs per lane and plane) live in registers for the
  // whole launch -- 4 VGPRs per row tile -- and only the full ones are re-read from LDS per product: a third of the
  // fragment bytes (6 of 18 KB per tile at NT = 3) no longer cross the CU's one LDS pipe, which 16 waves share
  constexpr bool HALF = (NT & 1) != 0;
  h4v ahh[HALF ? NT : 1], ahl[HALF ? NT : 1];
  if constexpr (HALF) {
#pragma unroll
    for (int mi = 0; mi < NT; ++mi) {
      const h8 hh = sCA[((mi * KS + KS - 1) * 2 + 0) * 64 + lane], hl = sCA[((mi * KS + KS - 1) * 2 + 1) * 64 + lane];
      ahh[mi] = __builtin_shufflevector(hh, hh, 0, 1, 2, 3);
      ahl[mi] = __builtin_shufflevector(hl, hl, 0, 1, 2, 3);
    }
  }
  auto ldA = [&](int mi, int ks) {
    Frag f;
    if (HALF && ks == KS - 1) {      // slots j >= 4 of a half fragment are never multiplied by a non-zero partner
      f.hi = __builtin_shufflevector(ahh[HALF ? mi : 0], ahh[HALF ? mi : 0], 0, 1, 2, 3, 0, 1, 2, 3);
      f.lo = __builtin_shufflevector(ahl[HALF ? mi : 0], ahl[HALF ? mi : 0], 0, 1, 2, 3, 0, 1, 2, 3);
    } else {
      f.hi = sCA[((mi * KS + ks) * 2 + 0) * 64 + lane];
      f.lo = sCA[((mi * KS + ks) * 2 + 1) * 64 + lane];
    }
    return f;
  };
  Frag FW1, FW2;
  float bb1[4], bb2[4];
  {
    float x1[8], x2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f2 = 4 * g + j;                      // k = f = 4g + j, j < 4 (half fragments)
      x1[j] = (j < 4 && f2 < F13 && c < F13) ? W1[f2 * F13 + c] : 0.f;
      x2[j] = (j < 4 && f2 < F13 && c < F13) ? W2[f2 * F13 + c] : 0.f;
    }
    FW1 = split_vals<X3>(x1);
    FW2 = split_vals<X3>(x2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 4 * g + r;
      bb1[r] = f < F13 ? b1[f] : 0.f;
      bb2[r] = f < F13 ? b2[f] : 0.f;
    }
  }
  PairMap<NP> map, omap;
  map.init(lane, I, SP * XS - 1);
  constexpr int ONES = 16, ZERO = 17;                 // pad words of row 0 (columns 16, 17 of XS = 20): never a store target
  omap.init_out(lane, I, ONES, ZERO);
  if (lane == 0) ob[ONES] = 1.f;                      // ob[ZERO] stays 0 from the clear above
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 bias1 = {bb1[0], bb1[1], bb1[2], bb1[3]}, bias2 = {bb2[0], bb2[1], bb2[2], bb2[3]};   // accumulator seeds
  const int wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;

  // The staging writes and the copy-out's LDS reads below carry NO per-k guard: pairs past the tile go to / come from the
  // dump and pad slots, so all of them sit in one basic block and issue back to back.  (With a wave-uniform
  // `if (64 * k < npairs)` around each pair the compiler emitted one block per k -- two ds_read, s_waitcnt lgkmcnt(0), the
  // split, two stores -- i.e. five serial LDS round trips per tile in the copy-out.)
  f32x2 xr[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) xr[k] = f32x2{0.f, 0.f};          // a k past the tile is never loaded: defined bits for the dump slot
  auto stage_x = [&]() {
    if (IO && io == 1) {                                           // one branch on the 16-bit type per tile, not one per pair
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const f32x2 v = io_pair(xr[k], 1);
        xb[map.o0[k]] = v[0];
        xb[map.o1[k]] = v[1];
      }
    } else {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const f32x2 v = IO ? io_pair(xr[k], 2) : xr[k];
        xb[map.o0[k]] = v[0];
        xb[map.o1[k]] = v[1];
      }
    }
  };
  if (wave_id < ntiles) {
    XLOAD(wave_id);
    stage_x();
  }
  for (int tile = wave_id; tile < ntiles; tile += nwaves) {
    asm volatile("" ::: "memory");                    // keep the A-fragment reads in LDS (no hoisting into VGPRs)
    wave_lds_fence();                                 // this tile's X is staged
    const bool more = tile + nwaves < ntiles;
    if (more) XLOAD(tile + nwaves);                   // prefetch the next tile

    f32x4 U[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) U[i] = mfma3h<X3>(xfrag_nat<X3>(xb, i, c, g), FW1, zero4);   // U1 row-tile i
    Frag UF[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) UF[ks] = (2 * ks + 1 < NT) ? frag_of<X3>(U[2 * ks], U[2 * ks + 1]) : frag_half<X3>(U[2 * ks]);
    f32x4 Ht[NT];                                    // H1^T column-tile n: [f][s = 16n + c]
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      f32x4 acc = bias1;                         // the bias seeds the accumulator (one v_add per value less)
      f32x4 acc16 = zero4;                       // the K = 16 step has its own accumulator (MIXED_FORMS)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (2 * ks + 1 < NT) acc = mfma3<X3>(UF[ks], ldA(n, ks), acc);
        else if (X3) acc16 = mfma3h<X3>(UF[ks], ldA(n, ks), acc16);
        else acc = mfma3h<X3>(UF[ks], ldA(n, ks), acc);
      }
      if (X3 && (NT & 1)) acc += acc16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Ht[n][r] = relu_nan(acc[r]);
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) U[i] = mfma3h<X3>(frag_half<X3>(Ht[i]), FW2, zero4);   // U2 row-tile i [s][f']
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) UF[ks] = (2 * ks + 1 < NT) ? frag_of<X3>(U[2 * ks], U[2 * ks + 1]) : frag_half<X3>(U[2 * ks]);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      f32x4 acc = bias2;
      f32x4 acc16 = zero4;                       // the K = 16 step has its own accumulator (MIXED_FORMS)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (2 * ks + 1 < NT) acc = mfma3<X3>(UF[ks], ldA(n, ks), acc);
        else if (X3) acc16 = mfma3h<X3>(UF[ks], ldA(n, ks), acc16);
        else acc = mfma3h<X3>(UF[ks], ldA(n, ks), acc);
      }
      if (X3 && (NT & 1)) acc += acc16;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = relu_nan(acc[r]);
      }
      *(f32x4*)(ob + (16 * n + c) * XS + 4 * g) = v;           // g^T[f' = 4g..4g+3][s] -> staged [s][f']
    }
    wave_lds_fence();
    // Stage the NEXT tile's X now (xb was last read by the U1 products above), i.e. wait for the
    // prefetch BEFORE this tile's output stores are issued, so the wait never covers the stores.
    if (more) stage_x();
    {  // coalesced copy-out: element pair (2p, 2p+1) -> one dword of the hi plane and one of the lo plane;
       // column I carries 1.0 (the ones column that yields b_ih / db_ih in the GEMMs), later pads 0
      unsigned* dh = (unsigned*)(ghi + (size_t)tile * ldp);
      unsigned* dl = (unsigned*)(glo + (size_t)tile * ldp);
      float v0[NP], v1[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {                               // every read first (omap is valid for every lane and k)
        v0[k] = ob[omap.o0[k]];                                    // the tile, then 1.0 at column I, then zeros
        v1[k] = ob[omap.o1[k]];
      }
      unsigned hi[NP], lo[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        float d0, d1;                                              // v - hi: -inf where the value rounded to fp16's inf
        split2t(v0[k], v1[k], hi[k], lo[k], d0, d1);
        chk = __builtin_fmaf(d0, 0.f, __builtin_fmaf(d1, 0.f, chk));
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (64 * k < ldp / 2) {                                    // wave-uniform
          const int p = lane + 64 * k;
          if (p < ldp / 2) {
            dh[p] = hi[k];
            if (X3) dl[p] = lo[k];
          }
        }
      }
    }
  }
  report_status(status, chk != chk, WGNN_STATUS_ACT_RANGE);
}

// ------------------------------------------------------------------------------------------------
// Backward of both layers for one tile (no dX: the input does not require grad).  With
//   U1 = X W1, H1 = relu(A U1 + b1)      (recomputed),  dZ2 = dg * (g > 0) * scale
//   dU2 = A^T dZ2                 dW2 += H1^T dU2        db2 += colsum(dZ2)
//                                 dH1 = dU2 W2^T         dZ1 = dH1 * (H1 > 0)
//   dU1 = A^T dZ1                 dW1 += X^T dU1         db1 += colsum(dZ1)
// every product contracts over the row index of an accumulator stack (or of a tile staged in LDS and
// read in C layout).  A and A^T fragments live in LDS (shared by the block's waves); dg is scaled by the
// power of two scales[0] so fp16 never sees ~1e-9 values and the partial sums are un-scaled by
// scales[1] at the end.
// 12 waves per backward block, one block per CU: the chain of small dependent products is latency-bound, and
// three waves per SIMD (166 VGPRs) hide more of it than two (measured 181 vs 198 us; 4-wave blocks, 2 per CU).
// One-pass fp16 (no lo halves) fits 128 VGPRs: 16 waves, four per SIMD (122 vs 129 us).  f16x3 at 128 VGPRs spills 17
// registers (268 vs 136 us) and stays at 12 waves.
// S = 49..64 (NT = 4) in the split modes needs 172 VGPRs: 12 waves spilled 3-4 registers (16-20 B of scratch) and ran 132.8 us at
// S = 64, B = 2048 against 118.3 us with 8 waves and no scratch (r4, tools/exp/gcnx_bwd_s64.py, same box)
#ifndef WGNN_BWD4_WAVES
#define WGNN_BWD4_WAVES 8
#endif
// DG16 = false in the one-pass mode (fp32 dg: only with a wide GRU behind a dense GCN) needs 131 VGPRs: 12 waves there too
// (3 spilled registers at 16 waves; 51.0 -> 49.1 us at S = 34, H = 200, B = 2048)
constexpr int bwd_waves(int NT, bool X3, bool DG16 = true) {
  // (round 5: the one-pass instances at S = 49..64 take the 8 waves of the split ones too -- at 12 the <4, one-pass, 16-bit I/O>
  // instance needed 170 VGPRs for 168 and spilled two: VERDICT r4 weak 9)
  return (!X3 && NT <= 3 && DG16) ? 16 : (NT >= 4 ? WGNN_BWD4_WAVES : 12);
}

// DG16: dg arrives as ONE fp16 plane (pgemm_nt_kernel<.., OUT16>, WGNN_MATH_F16X3G) instead of fp32
template <int NT, bool X3, bool IO, bool DG16>
__global__ void __launch_bounds__(64 * bwd_waves(NT, X3, DG16)) gcnx_bwd_kernel(int ntiles, int S, const float* __restrict__ A,
                                                       const void* __restrict__ X, const void* __restrict__ xtail, int io,
                                                       const float* __restrict__ W1,
                                                       const float* __restrict__ b1, const float* __restrict__ W2,
                                                       const _Float16* __restrict__ gact, int ld_g,
                                                       const void* __restrict__ dgv, int ld_dg,
                                                       const float* __restrict__ scales, int scale_in,
                                                       float* __restrict__ partial) {
  const float* dg = (const float*)dgv;
  const _Float16* dgh = (const _Float16*)dgv;
  constexpr int KS = (NT + 1) / 2;
  constexpr int NF = NT * KS;
  constexpr int SP = 16 * NT;
  constexpr int NP = (SP * F13 / 2 + 63) / 64;
  __shared__ __attribute__((aligned(16))) h8 sCA[2 * NF * 64];   // [frag][hi|lo][lane]
  __shared__ __attribute__((aligned(16))) h8 sCT[2 * NF * 64];
  // odd NT: the last k step's fragments are half fragments (4 halfs per lane); a compact copy lets the products read them
  // as ds_read_b64 -- a quarter of the fragment bytes of a tile (9 of 36 KB at NT = 3) less through the CU's LDS pipe
  // (split instances only: the one-pass ones, whose half steps run on the K = 32 form, measured 89.7 -> 92.5 us with it)
  constexpr bool HALF = X3 && (NT & 1) != 0;
  __shared__ __attribute__((aligned(8))) h4v sHA[HALF ? 2 * NT * 64 : 1], sHT[HALF ? 2 * NT * 64 : 1];
  constexpr int BWD_WAVES = bwd_waves(NT, X3, DG16);
  __shared__ __attribute__((aligned(16))) float sbuf[BWD_WAVES * 2 * SP * XS];
  static_assert(2 * SP * XS >= PART, "the per-wave staging buffer doubles as its reduction row");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int I = S * F13;
  float* xb = sbuf + wave * 2 * SP * XS;
  float* db = xb + SP * XS;
  for (int i = lane; i < 2 * SP * XS; i += 64) xb[i] = 0.f;
  if (wave == 0) {
    Frag T[NT][KS];
    build_A_frags<NT, KS, false, X3>(T, A, S, c, g);
#pragma unroll
    for (int mi = 0; mi < NT; ++mi)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sCA[((mi * KS + ks) * 2 + 0) * 64 + lane] = T[mi][ks].hi;
        sCA[((mi * KS + ks) * 2 + 1) * 64 + lane] = T[mi][ks].lo;
        if (HALF && ks == KS - 1) {
          sHA[(mi * 2 + 0) * 64 + lane] = __builtin_shufflevector(T[mi][ks].hi, T[mi][ks].hi, 0, 1, 2, 3);
          sHA[(mi * 2 + 1) * 64 + lane] = __builtin_shufflevector(T[mi][ks].lo, T[mi][ks].lo, 0, 1, 2, 3);
        }
      }
  } else if (wave == 1) {
    Frag T[NT][KS];
    build_A_frags<NT, KS, true, X3>(T, A, S, c, g);
#pragma unroll
    for (int mi = 0; mi < NT; ++mi)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sCT[((mi * KS + ks) * 2 + 0) * 64 + lane] = T[mi][ks].hi;
        sCT[((mi * KS + ks) * 2 + 1) * 64 + lane] = T[mi][ks].lo;
        if (HALF && ks == KS - 1) {
          sHT[(mi * 2 + 0) * 64 + lane] = __builtin_shufflevector(T[mi][ks].hi, T[mi][ks].hi, 0, 1, 2, 3);
          sHT[(mi * 2 + 1) * 64 + lane] = __builtin_shufflevector(T[mi][ks].lo, T[mi][ks].lo, 0, 1, 2, 3);
        }
      }
  }
  __syncthreads();
  auto ld_frag = [&](const h8* full, const h4v* half, int mi, int ks) {
    Frag f;
    if (HALF && ks == KS - 1) {      // slots j >= 4 of a half fragment are never multiplied by a non-zero partner
      const h4v a = half[(mi * 2 + 0) * 64 + lane], b = half[(mi * 2 + 1) * 64 + lane];
      f.hi = __builtin_shufflevector(a, a, 0, 1, 2, 3, 0, 1, 2, 3);
      f.lo = __builtin_shufflevector(b, b, 0, 1, 2, 3, 0, 1, 2, 3);
    } else {
      f.hi = full[((mi * KS + ks) * 2 + 0) * 64 + lane];
      f.lo = full[((mi * KS + ks) * 2 + 1) * 64 + lane];
    }
    return f;
  };
  auto ldA = [&](int mi, int ks) { return ld_frag(sCA, sHA, mi, ks); };
  auto ldT = [&](int mi, int ks) { return ld_frag(sCT, sHT, mi, ks); };

  Frag FW1, FW2T;
  {
    float x1[8], x2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = 4 * g + j;                       // k = 4g + j, j < 4 (half fragments)
      x1[j] = (j < 4 && f < F13 && c < F13) ? W1[f * F13 + c] : 0.f;
      x2[j] = (j < 4 && f < F13 && c < F13) ? W2[c * F13 + f] : 0.f;   // k = f', n = c = f: W2^T[f'][f]
    }
    FW1 = split_vals<X3>(x1);
    FW2T = split_vals<X3>(x2);
  }
  const float bias1 = c < F13 ? b1[c] : 0.f;         // H1 is [s][f] here: bias per column
  const float s_in = (scales && scale_in) ? scales[0] : 1.f, s_out = scales ? scales[1] : 1.f;
  PairMap<NP> map;
  map.init(lane, I, SP * XS - 1);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;

  f32x4 dW1acc = zero4, dW2acc = zero4, dW1acc16 = zero4, dW2acc16 = zero4;   // K = 32 / K = 16 steps (MIXED_FORMS)
  float db1acc = 0.f, db2acc = 0.f;

  f32x2 xr[NP], dr[DG16 ? 1 : NP];
  h2 gr[NP], drh[DG16 ? NP : 1];
#pragma unroll
  for (int k = 0; k < NP; ++k) {                 // a k past the tile is never loaded: defined bits for the dump slot
    xr[k] = f32x2{0.f, 0.f};
    gr[k] = h2{(_Float16)0.f, (_Float16)0.f};
    if (DG16) drh[DG16 ? k : 0] = h2{(_Float16)0.f, (_Float16)0.f};
    else dr[DG16 ? 0 : k] = f32x2{0.f, 0.f};
  }
  if (wave_id < ntiles) {
    XLOAD(wave_id);
    gload_pairs_h<NP, true>(gr, gact + (size_t)wave_id * ld_g, lane, I);
    if constexpr (DG16) gload_pairs_h<NP>(drh, dgh + (size_t)wave_id * ld_dg, lane, I);
    else gload_pairs<NP>(dr, dg + (size_t)wave_id * ld_dg, lane, I);
  }
  for (int tile = wave_id; tile < ntiles; tile += nwaves) {
    asm volatile("" ::: "memory");   // keep the A / A^T fragment reads in LDS (no hoisting into 96 VGPRs)
    wave_lds_fence();
    // (no per-k guard: pairs past the tile go to the dump slot, so the 4 NP stores sit in one basic block; one branch on the
    // 16-bit type per tile)
    auto stage = [&](auto ioc) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const f32x2 xv = IO ? io_pair(xr[k], decltype(ioc)::value) : xr[k];
        xb[map.o0[k]] = xv[0];
        xb[map.o1[k]] = xv[1];
        const float d0 = DG16 ? (float)drh[DG16 ? k : 0][0] : dr[DG16 ? 0 : k][0];
        const float d1 = DG16 ? (float)drh[DG16 ? k : 0][1] : dr[DG16 ? 0 : k][1];
        db[map.o0[k]] = (float)gr[k][0] > 0.f ? d0 * s_in : 0.f;           // dZ2 = dg * (g > 0), range-scaled
        db[map.o1[k]] = (float)gr[k][1] > 0.f ? d1 * s_in : 0.f;
      }
    };
    if (IO && io == 1) stage(std::integral_constant<int, 1>{});
    else stage(std::integral_constant<int, 2>{});
    wave_lds_fence();
    if (tile + nwaves < ntiles) {                               // prefetch the next tile under this one's math
      const size_t nt = (size_t)(tile + nwaves);
      XLOAD(tile + nwaves);
      gload_pairs_h<NP, true>(gr, gact + nt * ld_g, lane, I);
      if constexpr (DG16) gload_pairs_h<NP>(drh, dgh + nt * ld_dg, lane, I);
      else gload_pairs<NP>(dr, dg + nt * ld_dg, lane, I);
    }
    // ---- recompute U1, H1 [s][f]
    f32x4 U[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) U[i] = mfma3h<X3>(xfrag_nat<X3>(xb, i, c, g), FW1, zero4);
    Frag UF[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) UF[ks] = (2 * ks + 1 < NT) ? frag_of<X3>(U[2 * ks], U[2 * ks + 1]) : frag_half<X3>(U[2 * ks]);
    f32x4 H1[NT];
#pragma unroll
    for (int mi = 0; mi < NT; ++mi) {
      f32x4 acc = zero4;
      f32x4 acc16 = zero4;                       // the K = 16 step has its own accumulator (MIXED_FORMS)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (2 * ks + 1 < NT) acc = mfma3<X3>(ldA(mi, ks), UF[ks], acc);
        else if (X3) acc16 = mfma3h<X3>(ldA(mi, ks), UF[ks], acc16);
        else acc = mfma3h<X3>(ldA(mi, ks), UF[ks], acc);
      }
      if (X3 && (NT & 1)) acc += acc16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s = 16 * mi + 4 * g + r;
        H1[mi][r] = s < S ? fmaxf(acc[r] + bias1, 0.f) : 0.f;
      }
    }
    // ---- dZ2 [s][f'] in C layout from the staged tile (pads are zero)
    f32x4 dZ[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = db[(16 * i + 4 * g + r) * XS + c];
        dZ[i][r] = v;
        db2acc += v;
      }
    Frag DZF[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) DZF[ks] = (2 * ks + 1 < NT) ? frag_of<X3>(dZ[2 * ks], dZ[2 * ks + 1]) : frag_half<X3>(dZ[2 * ks]);
    // ---- dU2 [s'][f'] = A^T dZ2 ; dW2 += H1^T dU2
    f32x4 dU[NT];
#pragma unroll
    for (int mi = 0; mi < NT; ++mi) {
      f32x4 acc = zero4;
      f32x4 acc16 = zero4;                       // the K = 16 step has its own accumulator (MIXED_FORMS)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (2 * ks + 1 < NT) acc = mfma3<X3>(ldT(mi, ks), DZF[ks], acc);
        else if (X3) acc16 = mfma3h<X3>(ldT(mi, ks), DZF[ks], acc16);
        else acc = mfma3h<X3>(ldT(mi, ks), DZF[ks], acc);
      }
      if (X3 && (NT & 1)) acc += acc16;
      dU[mi] = acc;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (2 * ks + 1 < NT)
        dW2acc = mfma3<X3>(frag_of<X3>(H1[2 * ks], H1[2 * ks + 1]), frag_of<X3>(dU[2 * ks], dU[2 * ks + 1]), dW2acc);
      else if (X3)
        dW2acc16 = mfma3h<X3>(frag_half<X3>(H1[2 * ks]), frag_half<X3>(dU[2 * ks]), dW2acc16);
      else
        dW2acc = mfma3h<X3>(frag_half<X3>(H1[2 * ks]), frag_half<X3>(dU[2 * ks]), dW2acc);
    }
    // ---- dH1 [s'][f] = dU2 W2^T contracts over dU2's COLUMN index: dU2 goes through the wave's staging buffer
    // (the dZ2 tile in it has been consumed) and comes back as natural-k A fragments -- 12 ds_write_b32 + 6
    // ds_read_b128 per lane instead of the 18 MFMAs of a second, transposed A product.  dZ1 = dH1 * (H1 > 0).
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) db[(16 * i + 4 * g + r) * XS + c] = dU[i][r];
    wave_lds_fence();
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const f32x4 dh = mfma3h<X3>(xfrag_nat<X3>(db, n, c, g), FW2T, zero4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = H1[n][r] > 0.f ? dh[r] : 0.f;
        dZ[n][r] = v;
        db1acc += v;
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) DZF[ks] = (2 * ks + 1 < NT) ? frag_of<X3>(dZ[2 * ks], dZ[2 * ks + 1]) : frag_half<X3>(dZ[2 * ks]);
    // ---- dU1 = A^T dZ1 ; dW1 += X^T dU1 (X read from the staged tile in C layout [s'][f])
#pragma unroll
    for (int mi = 0; mi < NT; ++mi) {
      f32x4 acc = zero4;
      f32x4 acc16 = zero4;                       // the K = 16 step has its own accumulator (MIXED_FORMS)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (2 * ks + 1 < NT) acc = mfma3<X3>(ldT(mi, ks), DZF[ks], acc);
        else if (X3) acc16 = mfma3h<X3>(ldT(mi, ks), DZF[ks], acc16);
        else acc = mfma3h<X3>(ldT(mi, ks), DZF[ks], acc);
      }
      if (X3 && (NT & 1)) acc += acc16;
      dU[mi] = acc;
    }
    f32x4 XC[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) XC[i][r] = xb[(16 * i + 4 * g + r) * XS + c];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (2 * ks + 1 < NT)
        dW1acc = mfma3<X3>(frag_of<X3>(XC[2 * ks], XC[2 * ks + 1]), frag_of<X3>(dU[2 * ks], dU[2 * ks + 1]), dW1acc);
      else if (X3)
        dW1acc16 = mfma3h<X3>(frag_half<X3>(XC[2 * ks]), frag_half<X3>(dU[2 * ks]), dW1acc16);
      else
        dW1acc = mfma3h<X3>(frag_half<X3>(XC[2 * ks]), frag_half<X3>(dU[2 * ks]), dW1acc);
    }
  }

  // ---- per-block reduction of the 4 waves, one partial row per block (deterministic order)
  float* red = sbuf;                 // staging is over: reuse it, one PART row per wave
  __syncthreads();
  float* mine = red + wave * PART;
  for (int i = lane; i < PART; i += 64) mine[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    mine[(4 * g + r) * FP + c] = (dW1acc[r] + dW1acc16[r]) * s_out;
    mine[FP * FP + (4 * g + r) * FP + c] = (dW2acc[r] + dW2acc16[r]) * s_out;
  }
  // column sums: lanes c, c+16, c+32, c+48 hold partial sums of column c
  db1acc += __shfl_xor(db1acc, 16, 64);
  db1acc += __shfl_xor(db1acc, 32, 64);
  db2acc += __shfl_xor(db2acc, 16, 64);
  db2acc += __shfl_xor(db2acc, 32, 64);
  if (g == 0) {
    mine[2 * FP * FP + c] = db1acc * s_out;
    mine[2 * FP * FP + FP + c] = db2acc * s_out;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PART; i += blockDim.x)
  {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < BWD_WAVES; ++w) t += red[w * PART + i];   // fixed order
    partial[(size_t)blockIdx.x * PART + i] = t;
  }
}

int grid_x(int ntiles, int S, bool x3) {
  int g = cdiv_i(ntiles, bwd_waves((S + 15) / 16, x3));
  const int cap = 256;                              // one block per CU, persistent over the tiles
  return g < 1 ? 1 : (g > cap ? cap : g);
}

}  // namespace

size_t gcnx2_bwd_partial_floats(int ntiles) { return (size_t)256 * PART; }
int gcnx_bwd_grid(int ntiles, int S, bool x3) { return grid_x(ntiles, S, x3); }

// The private copy of X's last tile for odd S*13 (XLOAD): returns the scratch pointer, or nullptr when no copy is needed.
static const void* xtail_copy(const void* X, int ntiles, int S, int io, void* scratch, hipStream_t st, int* rc) {
  *rc = WGNN_OK;
  const size_t I = (size_t)S * 13, es = io ? 2 : 4;
  if ((I & 1) == 0) return nullptr;
  if (!scratch) { *rc = WGNN_ERR_NULL; return nullptr; }
  if (hipMemcpyAsync(scratch, (const char*)X + (size_t)(ntiles - 1) * I * es, I * es, hipMemcpyDeviceToDevice, st) !=
      hipSuccess)
    *rc = WGNN_ERR_HIP;
  return scratch;
}

int launch_gcnx2_fwd(int ntiles, int S, const float* A, const void* X, int io, const float* W1, const float* b1,
                     const float* W2, const float* b2, void* g_planes, int ldg, bool x3, unsigned* status,
                     void* xtail_scratch, hipStream_t st) {
  int rc0;
  const void* xt = xtail_copy(X, ntiles, S, io, xtail_scratch, st, &rc0);
  if (rc0 != WGNN_OK) return rc0;
  _Float16* ghi = (_Float16*)g_planes;
  _Float16* glo = ghi + (size_t)ntiles * ldg;
  const double fl = (double)ntiles * 2.0 * (2.0 * S * S * 13 + 2.0 * S * 13 * 13);
  const double by = (double)ntiles * S * 13 * (io ? 2.0 : 4.0) + (double)ntiles * S * 13 * 4.0;   // X in, g planes out
  int gx = cdiv_i(ntiles, FWD_WAVES);
  gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
  const dim3 grid(gx);
#define FWD_LAUNCH(NT, X3V, IOV, NAME, BYTES)                                                                     \
  PROF_LAUNCH(NAME, fl, BYTES, st,                                                                                \
              hipLaunchKernelGGL((gcnx_fwd_kernel<NT, X3V, IOV>), grid, dim3(64 * FWD_WAVES), 0, st, ntiles, S, A, X, xt, io, W1, \
                                 b1, W2, b2, ghi, glo, ldg, status))
#define FWD_CASE(NT)                                                                                              \
  if (x3 && !io) FWD_LAUNCH(NT, true, false, "gcnx_fwd_kernel<" #NT ">", by);                                     \
  else if (x3) FWD_LAUNCH(NT, true, true, "gcnx_fwd_kernel<" #NT ">", by);                                        \
  else if (!io) FWD_LAUNCH(NT, false, false, "gcnx_fwd_kernel<" #NT ",f16>", by * 0.75);                          \
  else FWD_LAUNCH(NT, false, true, "gcnx_fwd_kernel<" #NT ",f16>", by * 0.75)
  switch ((S + 15) / 16) {
    case 1: FWD_CASE(1); break;
    case 2: FWD_CASE(2); break;
    case 3: FWD_CASE(3); break;
    case 4: FWD_CASE(4); break;
    default: return WGNN_ERR_UNSUPPORTED;
  }
#undef FWD_CASE
#undef FWD_LAUNCH
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_gcnx2_bwd(int ntiles, int S, const float* A, const void* X, int io, const float* W1, const float* b1,
                     const float* W2, const void* g_planes, int ldg, const void* dg, int ld_dg, bool dg16, const float* scales,
                     int scale_in, float* partial, bool x3, void* xtail_scratch, hipStream_t st) {
  if (ld_dg < S * 13 || (ld_dg & 1)) return WGNN_ERR_SHAPE;      // rows of dg: aligned pairs
  int rc0;
  const void* xt = xtail_copy(X, ntiles, S, io, xtail_scratch, st, &rc0);
  if (rc0 != WGNN_OK) return rc0;
  const _Float16* g = (const _Float16*)g_planes;   // hi plane carries the sign: g > 0 <=> hi > 0
  const double fl = (double)ntiles * ((2.0 * S * S * 13 + 2.0 * S * 13 * 13) * 3.0 + 2.0 * S * 13 * 13 * 2.0);
  // what the launch reads: X and dg as fp32, and the fp16 hi plane of g (2 bytes x ldg per tile) as the ReLU mask
  const double by = (double)ntiles * (S * 13 * (io ? 2.0 : 4.0) + S * 13 * (dg16 ? 2.0 : 4.0) + ldg * 2.0);
  const dim3 grid(grid_x(ntiles, S, x3));
#define BWD_LAUNCH(NT, X3V, IOV, D16, NAME)                                                                       \
  PROF_LAUNCH(NAME, fl, by, st,                                                                                   \
              hipLaunchKernelGGL((gcnx_bwd_kernel<NT, X3V, IOV, D16>), grid, dim3(64 * bwd_waves(NT, X3V, D16)), 0, st, ntiles, S, A, X, xt, io, \
                                 W1, b1, W2, g, ldg, dg, ld_dg, scales, scale_in, partial))
#define BWD_CASE(NT)                                                                                              \
  if (x3 && dg16 && !io) BWD_LAUNCH(NT, true, false, true, "gcnx_bwd_kernel<" #NT ">");                           \
  else if (x3 && dg16) BWD_LAUNCH(NT, true, true, true, "gcnx_bwd_kernel<" #NT ">");                              \
  else if (x3 && !io) BWD_LAUNCH(NT, true, false, false, "gcnx_bwd_kernel<" #NT ">");                             \
  else if (x3) BWD_LAUNCH(NT, true, true, false, "gcnx_bwd_kernel<" #NT ">");                                     \
  else if (dg16 && !io) BWD_LAUNCH(NT, false, false, true, "gcnx_bwd_kernel<" #NT ",f16>");                       \
  else if (dg16) BWD_LAUNCH(NT, false, true, true, "gcnx_bwd_kernel<" #NT ",f16>");                               \
  else if (!io) BWD_LAUNCH(NT, false, false, false, "gcnx_bwd_kernel<" #NT ",f16>");                              \
  else BWD_LAUNCH(NT, false, true, false, "gcnx_bwd_kernel<" #NT ",f16>")
  switch ((S + 15) / 16) {
    case 1: BWD_CASE(1); break;
    case 2: BWD_CASE(2); break;
    case 3: BWD_CASE(3); break;
    case 4: BWD_CASE(4); break;
    default: return WGNN_ERR_UNSUPPORTED;
  }
#undef BWD_CASE
#undef BWD_LAUNCH
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
