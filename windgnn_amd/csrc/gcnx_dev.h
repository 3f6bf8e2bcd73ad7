// Device helpers shared by the split-fp16 GCN kernels (gcnx.hip) and the fused GCN -> input-projection forward (gcngi.hip):
// fragment construction (hi/lo splits in registers), the half-depth / full-depth MFMA wrappers, per-wave LDS staging of a
// [S][13] tile and the A-matrix fragment builder.  See gcnx.hip's header comment for the register-chaining scheme.
#pragma once
#include "common.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int F13 = 13;
constexpr int FP = 16;
constexpr int PART = 2 * FP * FP + 2 * FP;   // same partial layout as gcn.hip

struct Frag { h8 hi, lo; };

__device__ __forceinline__ f32x4 mfma_x(h8 a, h8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x16_f16 on the first four k slots (j < 4) of two fragments: half the matrix-pipe time of the K = 32
// form.  Used wherever the slots j >= 4 are padding: every product over the 13 -> 16 features, and the last k step of a
// product over stations when the number of 16-row tiles is odd (S = 34: stations 32..47 of 32..63).
// MIXED_FORMS: the two forms never feed one accumulator -- a product that has steps of both kinds keeps one accumulator
// per form and adds them on the VALU.  hipcc (ROCm 7.2) does not insert the wait states between a v_mfma_f32_16x16x32_f16
// and a v_mfma_f32_16x16x16_f16 of which one takes the other's result as SrcC: a two-MFMA test kernel read a stale
// accumulator (right again with any instruction in between), and so did these kernels in either order of the steps.
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_h(h8 a, h8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(a, a, 0, 1, 2, 3),
                                               __builtin_shufflevector(b, b, 0, 1, 2, 3), c, 0, 0, 0);
}
// X3 = true: split-fp16 product lo*hi + hi*lo + hi*hi (fp32-grade); X3 = false: plain fp16 operands, one pass
// (the "f16" math mode for BASELINE's 16-bit configuration; lo halves are never formed).
template <bool X3>
__device__ __forceinline__ f32x4 mfma3(const Frag& a, const Frag& b, f32x4 c) {
  if (X3) {
    c = mfma_x(a.lo, b.hi, c);
    c = mfma_x(a.hi, b.lo, c);
  }
  c = mfma_x(a.hi, b.hi, c);
  return c;
}
// (one-pass fp16 mode: the K = 32 form on the same half fragments, whose slots j >= 4 are zero, into the product's one
// accumulator -- that instance is bound by its dependent chain, not by VALU issue, and the mixed forms with their extra
// accumulators and adds only cost it registers; same-box A/B against the code before the half fragments: 113 vs 111 us)
template <bool X3>
__device__ __forceinline__ f32x4 mfma3h(const Frag& a, const Frag& b, f32x4 c) {
  if (X3) {
    c = mfma_h(a.lo, b.hi, c);
    c = mfma_h(a.hi, b.lo, c);
    c = mfma_h(a.hi, b.hi, c);
  } else {
    c = mfma_x(a.hi, b.hi, c);
  }
  return c;
}
// hi = fp16(x) (v_cvt_pk_f16_f32, 2 values per instruction), lo = fp16(x - hi) with the difference
// formed by v_fma_mix_f32 reading hi straight out of the packed register: 2 VALU per value instead
// of the 3 hipcc emits for the C expression (these kernels are VALU-issue-bound on exactly this).
// The hi conversion stays a compiler-visible instruction: x is usually an MFMA result, and the
// XDL-write -> VALU-read wait states are only inserted for instructions the hazard recognizer can
// see (an asm block reading a VGPR-form MFMA result directly gets none and reads stale registers).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt2(float x0, float x1) {
  const f32x2 xv = {x0, x1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(xv, h2));
}
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  float t0, t1;
  hi = cvt2(x0, x1);
  // only the two v_fma_mix_f32 (no C expression selects them) are asm; the conversion of their results is a
  // compiler-visible v_cvt_pk_f16_f32 again, so the VALU-write -> MFMA-operand wait states are the hazard
  // recognizer's business (it fills them with independent instructions; an `s_nop 1` inside the asm could not be
  // scheduled around: 82 of the 713 instructions of a backward tile were such nops)
  asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(t0), "=&v"(t1)
      : "v"(hi), "v"(x0), "v"(x1));
  lo = cvt2(t0, t1);
}
// the same split, also handing back the differences x - hi (for the copy-out's range check: -inf / NaN where x left fp16's range)
__device__ __forceinline__ void split2t(float x0, float x1, unsigned& hi, unsigned& lo, float& t0, float& t1) {
  hi = cvt2(x0, x1);
  asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(t0), "=&v"(t1)
      : "v"(hi), "v"(x0), "v"(x1));
  lo = cvt2(t0, t1);
}
template <bool X3>
__device__ __forceinline__ Frag split_vals(const float (&x)[8]) {
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  u32x4v hi, lo = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h, l = 0u;
    if (X3) split2(x[2 * j], x[2 * j + 1], h, l);
    else h = cvt2(x[2 * j], x[2 * j + 1]);
    hi[j] = h;
    lo[j] = l;
  }
  Frag f;
  f.hi = __builtin_bit_cast(h8, hi);
  f.lo = __builtin_bit_cast(h8, lo);
  return f;
}
// operand fragment from two stacked accumulator row-tiles (slots j<4 from t0, j>=4 from t1)
template <bool X3>
__device__ __forceinline__ Frag frag_of(f32x4 t0, f32x4 t1) {
  const float x[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
  return split_vals<X3>(x);
}
// half fragment: slots j < 4 from one accumulator row-tile (or four values), slots j >= 4 unused (mfma3h)
template <bool X3>
__device__ __forceinline__ Frag frag_half(f32x4 t) {
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  u32x4v hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    unsigned h, l = 0u;
    if (X3) split2(t[2 * j], t[2 * j + 1], h, l);
    else h = cvt2(t[2 * j], t[2 * j + 1]);
    hi[j] = h;
    lo[j] = l;
  }
  Frag f;
  f.hi = __builtin_bit_cast(h8, hi);
  f.lo = __builtin_bit_cast(h8, lo);
  return f;
}
// max(z, 0) for finite z; NaN for z = NaN or +-inf (0 * z is +-0 or NaN): one v_fma more than the plain ReLU
__device__ __forceinline__ float relu_nan(float z) { return __builtin_fmaf(z, 0.f, fmaxf(z, 0.f)); }
__device__ __forceinline__ int rho(int ks, int g, int j) { return 16 * (2 * ks + (j >> 2)) + 4 * g + (j & 3); }

// "A [m = s][k = s']" fragments in rho order (also the B operand of any product with A^T)
template <int NT, int KS, bool TRANSPOSE, bool X3>
__device__ __forceinline__ void build_A_frags(Frag (&CA)[NT][KS], const float* __restrict__ A, int S, int c, int g) {
#pragma unroll
  for (int mi = 0; mi < NT; ++mi)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int m = 16 * mi + c, k = rho(ks, g, j);
        float v = 0.f;
        if (m < S && k < S) v = TRANSPOSE ? A[k * S + m] : A[m * S + k];
        x[j] = v;
      }
      CA[mi][ks] = split_vals<X3>(x);
    }
}

// ------------------------------------------------------------------------------------------------
// Per-wave LDS staging: a tile's S*13 floats are moved HBM <-> LDS with fully coalesced dword
// accesses (element i = lane + 64k) and laid out [station][XS] so that fragment reads are aligned
// ds_read_b128 (natural k = f) or ds_read_b32 (C layout).  Each wave owns its buffers; ordering
// between its own LDS writes and reads needs no s_barrier, only a fence the compiler respects.
constexpr int XS = 20;   // row stride in floats: 16-B aligned rows, spreads rows over banks

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Linear tile element pairs (2p, 2p+1), p = lane + 64k, <-> LDS offsets s*XS + f.  Tiles start 8-byte
// aligned in HBM (S*13*4 bytes per tile), so every global access is a coalesced 8-byte (fp32) or
// 4-byte (fp16 pair) one; iterations whose 64 pairs all lie past the tile are skipped by a scalar branch.
template <int NP>
struct PairMap {
  int o0[NP], o1[NP];
  // elements past the tile map to `dump`, a pad slot no fragment read touches: every LDS store is
  // then unconditional (no exec-mask branch per element)
  __device__ __forceinline__ void init(int lane, int I, int dump) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int e = 2 * (lane + 64 * k);
      o0[k] = e < I ? (e / F13) * XS + (e % F13) : dump;
      o1[k] = e + 1 < I ? ((e + 1) / F13) * XS + ((e + 1) % F13) : dump;
    }
  }
  // read map of the forward's copy-out: element I of the padded row reads a slot that holds 1.0 (the ones column of the
  // g planes), later elements a slot that holds 0 -- two pad words of row 0 no tile store touches -- so the copy-out needs
  // no selects (4 v_cndmask per pair before)
  __device__ __forceinline__ void init_out(int lane, int I, int ones, int zero) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int e = 2 * (lane + 64 * k);
      o0[k] = e < I ? (e / F13) * XS + (e % F13) : (e == I ? ones : zero);
      o1[k] = e + 1 < I ? ((e + 1) / F13) * XS + ((e + 1) % F13) : (e + 1 == I ? ones : zero);
    }
  }
};

// STREAM: the tensor is read once per launch and will not be reused before the cache has turned over (X, the old g plane):
// non-temporal loads do not allocate in the 256 MB Infinity Cache, so they do not evict -- and wait for the write-back of --
// the dirty lines the previous kernels left there.  (tools/exp/stream_rows.hip: a 174 MB read behind 768 MB of dirty lines
// takes 61 us with plain loads and 28 us with nt loads.)  Only for whole-line, read-once streams of OLD data: operands a
// neighbouring kernel has just written (dg here, GI and the gate stash in the recurrences) are still in the cache as dirty
// lines and a non-temporal load of those is slow (grux_fwd 88 -> 101 us with nt GI loads), and streams read in 64-byte
// pieces by several instructions (the recurrences' labels) lose their L2 reuse (87 -> 90.5 us): both measured, both plain.
template <int NP, bool STREAM = false>
__device__ __forceinline__ void gload_pairs(f32x2 (&r)[NP], const float* __restrict__ src, int lane, int I) {
  const int npairs = (I + 1) / 2;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (64 * k < npairs) {                               // wave-uniform
      const int p = lane + 64 * k;
      const f32x2* q = (const f32x2*)(src + 2 * (p < npairs ? p : npairs - 1));   // clamped: no exec-mask branches
      r[k] = STREAM ? __builtin_nontemporal_load(q) : *q;
    }
  }
}
// X of tile t.  With an odd tile length the last pair of a tile ends one element past it -- harmless (the next tile's
// first element, dropped at staging) except at the tensor's LAST tile, where it would be a read past a caller-owned
// buffer: the launcher copies that tile into scratch with one element of slack (xtail; nullptr when S*13 is even) and
// the wave that owns it reads the copy.  A scalar select of the base address: no VGPRs, nothing in the common path.
#define XLOAD(t)                                                                                          \
  do {                                                                                                    \
    const bool tl_ = xtail != nullptr && (t) == ntiles - 1;                                               \
    gload_pairs_io<NP>(xr, tl_ ? xtail : X, tl_ ? (size_t)0 : (size_t)(t) * I, lane, I, IO ? io : 0);     \
  } while (0)
// The same for an io-typed tensor (wgnn_io): 16-bit pairs are one dword, kept raw in r[k][0] until io_pair() converts
// them at staging time (so the wait for the prefetch stays where it was).  `tile_elems` = element offset of the tile.
template <int NP>
__device__ __forceinline__ void gload_pairs_io(f32x2 (&r)[NP], const void* __restrict__ base, size_t tile_elems,
                                               int lane, int I, int io) {   // X: always a stream
  if (io == 0) {
    gload_pairs<NP, true>(r, (const float*)base + tile_elems, lane, I);
    return;
  }
  const unsigned short* src = (const unsigned short*)base + tile_elems;
  const int npairs = (I + 1) / 2;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (64 * k < npairs) {
      const int p = lane + 64 * k;
      r[k][0] = __builtin_bit_cast(float, __builtin_nontemporal_load((const unsigned*)(src + 2 * (p < npairs ? p : npairs - 1))));
    }
  }
}
__device__ __forceinline__ f32x2 io_pair(f32x2 raw, int io) {
  if (io == 0) return raw;
  const unsigned u = __builtin_bit_cast(unsigned, raw[0]);
  f32x2 v;
  if (io == 1) {
    v = __builtin_convertvector(__builtin_bit_cast(h2, u), f32x2);
  } else {   // bf16: the upper half of an fp32
    v[0] = __builtin_bit_cast(float, u << 16);
    v[1] = __builtin_bit_cast(float, u & 0xffff0000u);
  }
  return v;
}
template <int NP, bool STREAM = false>
__device__ __forceinline__ void gload_pairs_h(h2 (&r)[NP], const _Float16* __restrict__ src, int lane, int I) {
  const int npairs = (I + 1) / 2;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (64 * k < npairs) {
      const int p = lane + 64 * k;
      const h2* q = (const h2*)(src + 2 * (p < npairs ? p : npairs - 1));
      r[k] = STREAM ? __builtin_nontemporal_load(q) : *q;
    }
  }
}

// A-operand half fragment (k = f = 4g + j, j < 4) of rows 16i + c of a staged [s][XS] tile
template <bool X3>
__device__ __forceinline__ Frag xfrag_nat(const float* xb, int i, int c, int g) {
  return frag_half<X3>(*(const f32x4*)(xb + (16 * i + c) * XS + 4 * g));
}

}  // namespace
