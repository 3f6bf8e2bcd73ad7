// The tail of the training step as ONE launch (src/main.py:79-80: the end of loss.backward() and optimizer.step()):
//   * fixed-order reduction of the split-K partials of the two GRU weight-gradient products (dW_ih|db_ih, dW_hh|db_hh)
//     and of the per-workgroup partials of the GCN backward (dW1, db1, dW2, db2) into the 8 gradients,
//   * torch.optim.Adam (defaults semantics) on the 8 parameters,
//   * the "prepared" images of W_ih the next step's GEMMs stage (fp16 hi/lo stage-major planes of [W_ih | b_ih] and of
//     W_ih^T in the fp16-plane modes, zero-padded fp32 copies in exact fp32) -- Adam already holds the new weights.
// It replaces, per step, tn_reduce x2 (or splitk_reduce x2), gcn_partial_reduce, adam and next step's split_weight2 x2
// (or pad_weight x2): six launch-sized passes.  With a gradient all-reduce in between (N > 1) the same kernel runs as
// reduce-only launches (GRU part early, conv part late) and one Adam + prepare launch that reads the summed bucket.
// Every sum has a fixed order: results are bitwise reproducible and identical between the fused and the split form.
#include "common.h"
#include <type_traits>

namespace {

constexpr int FP = 16, PART = 2 * FP * FP + 2 * FP;   // GCN partial row: dW1 | dW2 | db1 | db2 (gcn.hip)
enum { T_C1W = 0, T_C1B, T_C2W, T_C2B, T_WIH, T_WHH, T_BIH, T_BHH };

// IMG = false: the caller writes the prepared images itself (seg_tn's wide stores); *wout receives the new weight
template <int ADAM, bool IMG = true>   // ADAM 1: Adam + prepared images after the reduction; 0: reduce only
__device__ __forceinline__ void emit(const FinishArgs& a, int t, int64_t idx, int row, int col, float gval, bool write_g,
                                     bool& bad_g, bool& bad_w, float* wout = nullptr) {
  bad_g |= !(__builtin_fabsf(gval) <= 3.0e38f);                       // inf / NaN in a final gradient
  if (write_g) a.g[t][idx] = gval;
  if (!ADAM) return;
  // torch.optim.Adam, single-tensor formulas in fp32, explicit roundings (the fused and the split launch agree bitwise)
  const float mi = __fmaf_rn(a.b1, a.m[t][idx], __fmul_rn(1.f - a.b1, gval));
  const float vi = __fmaf_rn(a.b2, a.v[t][idx], __fmul_rn(__fmul_rn(1.f - a.b2, gval), gval));
  a.m[t][idx] = mi;
  a.v[t][idx] = vi;
  const float denom = __fadd_rn(__fmul_rn(__fsqrt_rn(vi), a.inv_sqrt_bc2), a.eps);
  const float w = __fsub_rn(a.p[t][idx], __fmul_rn(a.lr_over_bc1, __fdiv_rn(mi, denom)));
  a.p[t][idx] = w;
  if (!IMG) { *wout = w; return; }
  if (a.prep_kind == 0 || (t != T_WIH && t != T_BIH)) return;
  if (t == T_BIH) { col = a.I; row = (int)idx; }                           // b_ih rides in column I of the forward image
  if (a.prep_kind == 1) {
    bad_w |= out_of_fp16_range(w);
    float ws = w;
    asm volatile("" : "+v"(ws));          // split the stored fp32 value (no fp16-output fma contraction: see seg_tn)
    const _Float16 h = (_Float16)ws, l = (_Float16)(ws - (float)h);
    const size_t f = bimg_off(row, col, a.np_g3);
    a.pf_hi[f] = h;
    a.pf_lo[f] = l;
    if (t == T_WIH) {
      const size_t b = bimg_off(col, row, a.np_i);
      a.pb_hi[b] = h;
      a.pb_lo[b] = l;
    }
  } else {
    a.wp[(size_t)row * a.Ip + col] = w;
    if (t == T_WIH) a.wt[(size_t)col * a.Gp + row] = w;
  }
}

// ---- split-K partials in pgemm_tn_kernel's own layout [z][tile][wave][i][j][lane][4] (pgemm.hip)
template <int ADAM>   // 1: Adam + prepared images after the reduction; 0: reduce only
__device__ __forceinline__ void seg_tn(const FinishArgs& a, const FinSeg& s, int tw, int tb, int blk, bool& bad_g,
                                       bool& bad_w) {
  const size_t slab4 = (size_t)s.ntiles * TN_WAVES * 5 * s.T * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;            // 64 quads x 4 z phases per block
  f32x4 v;
  size_t q;
  if (s.splitk < 4) {
    // few K chunks (huge weight matrices: BASELINE configs[4] has ONE): a quad per thread, every lane busy -- with the 4
    // z phases below three quarters of the block would idle through the loads (2.4 G elements: 36 ms of a 155 ms step)
    q = (size_t)blk * 256 + threadIdx.x;
    if (q >= slab4) return;
    const f32x4* src = (const f32x4*)s.partial + q;
    v = src[0];
    for (int z = 1; z < s.splitk; ++z) v += src[(size_t)z * slab4];
  } else {
    q = (size_t)blk * 64 + tx;
    const f32x4* src = (const f32x4*)s.partial + q;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int z = ty;
    for (; z + 12 < s.splitk; z += 16) {                               // 4 independent 16-byte loads in flight
      s0 += src[(size_t)z * slab4];
      s1 += src[(size_t)(z + 4) * slab4];
      s2 += src[(size_t)(z + 8) * slab4];
      s3 += src[(size_t)(z + 12) * slab4];
    }
    for (; z < s.splitk; z += 4) s0 += src[(size_t)z * slab4];
    __shared__ f32x4 red[4][64];
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty != 0) return;
    v = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  }
  if (s.scaled) v *= a.scales[1];
  size_t t = q;
  const int lane = (int)(t % 64); t /= 64;
  const int j = (int)(t % s.T); t /= s.T;
  const int i = (int)(t % 5); t /= 5;
  const int wave = (int)(t % TN_WAVES);
  const int tile = (int)(t / TN_WAVES);
  const int mb = tile / s.nNb, nb = tile % s.nNb, wm = wave & 3, wn = wave >> 2;
  const int n = nb * 32 * s.T + 16 * (s.T * wn + j) + (lane & 15);
  const int m0 = mb * TN_BM + 80 * wm + 16 * i + 4 * (lane >> 4);
  if (n >= s.Nout) return;
  // Wide image stores (W_ih with fp16-plane images, dimensions that are multiples of 4 -- BASELINE configs[4]: 36864 x 53248):
  // the thread's 4 consecutive rows of one column are 4 consecutive halfs of the W_ih^T image (one 8-byte store per plane),
  // and a 4 x 4 transpose inside the lane quad (two DPP exchanges) turns its 4 rows x the quad's 4 columns into one row x 4
  // consecutive columns of the forward image (one 8-byte store per plane) -- instead of sixteen 2-byte stores per thread, which
  // ran the 15.8 GB of image writes of that configuration at 1.3 TB/s (12.4 of the finish launch's 32.8 ms)
  if (ADAM && tw == T_WIH && a.prep_kind == 1 && s.msplit == 0 && (s.ncols & 3) == 0 && (s.Mout & 3) == 0 && n < s.ncols &&
      m0 < s.Mout) {
    unsigned pk[4];                                                    // hi | lo << 16 of the new weights, rows m0 .. m0 + 3
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float w;
      emit<ADAM, false>(a, tw, (int64_t)(m0 + r) * s.ncols + n, m0 + r, n, v[r], true, bad_g, bad_w, &w);
      // the image is the split of the STORED fp32 weight: without the barrier hipcc contracts "fp16(p - lr q)" into ONE
      // v_fma_mixlo_f16 (a single rounding of the exact value), which differs from fp16(fp32(..)) at exact ties -- the
      // image would no longer be bit-identical to wgnn_prepare_weights of the parameters (4 of 156 000 elements in the test)
      asm volatile("" : "+v"(w));
      bad_w |= out_of_fp16_range(w);
      const _Float16 h = (_Float16)w, l = (_Float16)(w - (float)h);
      pk[r] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
    }
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    {  // W_ih^T image: B row = column n of W_ih, k = W_ih row m0 .. m0 + 3 (4 consecutive halfs of one 16-byte fragment piece)
      const size_t b = bimg_off(n, m0, a.np_i);
      us4 hh, ll;
#pragma unroll
      for (int r = 0; r < 4; ++r) { hh[r] = (unsigned short)(pk[r] & 0xffffu); ll[r] = (unsigned short)(pk[r] >> 16); }
      *(us4*)(a.pb_hi + b) = hh;
      *(us4*)(a.pb_lo + b) = ll;
    }
    {  // forward image: B row = W_ih row, k = 4 consecutive columns (bimg_off); quad transpose: lane L of the quad ends with row m0 + L
      const int L = lane & 3;
      const bool odd = L & 1, upper = L & 2;
      auto xch = [](unsigned x, auto ctrl) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, 0xF, 0xF, false);
      };
      using X1 = std::integral_constant<int, 0xB1>;                    // quad_perm [1,0,3,2]: lane ^ 1
      using X2 = std::integral_constant<int, 0x4E>;                    // quad_perm [2,3,0,1]: lane ^ 2
      // step 1 (lane ^ 1): even lanes end with rows {0, 2} x columns {c, c + 1}, odd lanes with rows {1, 3} x {c - 1, c}
      const unsigned r0 = xch(odd ? pk[0] : pk[1], X1{}), r1 = xch(odd ? pk[2] : pk[3], X1{});
      const unsigned b0 = odd ? r0 : pk[0], b1 = odd ? pk[1] : r0, b2 = odd ? r1 : pk[2], b3 = odd ? pk[3] : r1;
      // step 2 (lane ^ 2): lanes 0, 1 keep their first row (columns 0, 1) and receive its columns 2, 3; lanes 2, 3 the second
      const unsigned s0 = xch(upper ? b0 : b2, X2{}), s1 = xch(upper ? b1 : b3, X2{});
      const unsigned c0 = upper ? s0 : b0, c1 = upper ? s1 : b1, c2 = upper ? b2 : s0, c3 = upper ? b3 : s1;
      const int row = m0 + L, col = n - L;                             // this lane's row, the quad's first column
      const size_t f = bimg_off(row, col, a.np_g3);
      us4 hh = {(unsigned short)(c0 & 0xffffu), (unsigned short)(c1 & 0xffffu), (unsigned short)(c2 & 0xffffu), (unsigned short)(c3 & 0xffffu)};
      us4 ll = {(unsigned short)(c0 >> 16), (unsigned short)(c1 >> 16), (unsigned short)(c2 >> 16), (unsigned short)(c3 >> 16)};
      *(us4*)(a.pf_hi + f) = hh;
      *(us4*)(a.pf_lo + f) = ll;
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int m = m0 + r;
    if (s.msplit > 0) {                                               // two-source A operand: GEMM rows -> weight rows
      if (m >= s.msplit) m = s.rows1 + (m - s.msplit);
      else if (m >= s.rows1) continue;
    }
    if (m >= s.Mout) continue;
    if (n < s.ncols) emit<ADAM>(a, tw, (int64_t)m * s.ncols + n, m, n, v[r], true, bad_g, bad_w);
    else if (n == s.Nout - 1) emit<ADAM>(a, tb, m, m, 0, v[r], true, bad_g, bad_w);
  }
}

// ---- split-K partials as plain [z][Mout][Nout] (gemm.hip, gemm32.hip)
template <int ADAM>   // 1: Adam + prepared images after the reduction; 0: reduce only
__device__ __forceinline__ void seg_plain(const FinishArgs& a, const FinSeg& s, int tw, int tb, int blk, bool& bad_g,
                                          bool& bad_w) {
  const size_t MN = (size_t)s.Mgemm * s.pitch;                       // Mgemm: rows of the GEMM that wrote the partials
  const size_t i = (size_t)blk * 256 + threadIdx.x;
  if (i >= MN) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  int z = 0;
  for (; z + 8 <= s.splitk; z += 8) {                                // 8 independent loads in flight; fixed order
    s0 += s.partial[(size_t)z * MN + i];
    s1 += s.partial[(size_t)(z + 1) * MN + i];
    s2 += s.partial[(size_t)(z + 2) * MN + i];
    s3 += s.partial[(size_t)(z + 3) * MN + i];
    s4 += s.partial[(size_t)(z + 4) * MN + i];
    s5 += s.partial[(size_t)(z + 5) * MN + i];
    s6 += s.partial[(size_t)(z + 6) * MN + i];
    s7 += s.partial[(size_t)(z + 7) * MN + i];
  }
  for (; z < s.splitk; ++z) s0 += s.partial[(size_t)z * MN + i];
  float v = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  if (s.scaled) v *= a.scales[1];
  int m = (int)(i / s.pitch);
  const int n = (int)(i % s.pitch);
  if (n >= s.Nout) return;                                            // alignment padding of the partial rows
  if (s.msplit > 0) {                                                 // two-source A operand: GEMM rows -> weight rows
    if (m >= s.msplit) m = s.rows1 + (m - s.msplit);
    else if (m >= s.rows1) return;
  }
  if (m >= s.Mout) return;
  if (n < s.ncols) emit<ADAM>(a, tw, (int64_t)m * s.ncols + n, m, n, v, true, bad_g, bad_w);
  else if (n == s.Nout - 1) emit<ADAM>(a, tb, m, m, 0, v, true, bad_g, bad_w);
}

// ---- per-workgroup partial rows of the GCN backward, [rows][PART]: 32 columns x 8 row groups per block
template <int ADAM>   // 1: Adam + prepared images after the reduction; 0: reduce only
__device__ __forceinline__ void seg_conv(const FinishArgs& a, int blk, bool& bad_g, bool& bad_w) {
  __shared__ float sm[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i = blk * 32 + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < PART) {
    int b = ty;
    for (; b + 24 < a.conv_rows; b += 32) {
      s0 += a.conv_partial[(size_t)b * PART + i];
      s1 += a.conv_partial[(size_t)(b + 8) * PART + i];
      s2 += a.conv_partial[(size_t)(b + 16) * PART + i];
      s3 += a.conv_partial[(size_t)(b + 24) * PART + i];
    }
    for (; b < a.conv_rows; b += 8) s0 += a.conv_partial[(size_t)b * PART + i];
  }
  sm[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty != 0 || i >= PART) return;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += sm[k][tx];
  constexpr int F = 13;
  if (i < FP * FP) {
    const int r = i / FP, c = i % FP;
    if (r < F && c < F) emit<ADAM>(a, T_C1W, r * F + c, r, c, s, true, bad_g, bad_w);
  } else if (i < 2 * FP * FP) {
    const int j = i - FP * FP, r = j / FP, c = j % FP;
    if (r < F && c < F) emit<ADAM>(a, T_C2W, r * F + c, r, c, s, true, bad_g, bad_w);
  } else if (i < 2 * FP * FP + FP) {
    const int c = i - 2 * FP * FP;
    if (c < F) emit<ADAM>(a, T_C1B, c, c, 0, s, true, bad_g, bad_w);
  } else {
    const int c = i - 2 * FP * FP - FP;
    if (c < F) emit<ADAM>(a, T_C2B, c, c, 0, s, true, bad_g, bad_w);
  }
}

// ---- tensors whose gradient is already final in g (after an all-reduce, or written directly by their kernel)
template <int ADAM>   // 1: Adam + prepared images after the reduction; 0: reduce only
__device__ __forceinline__ void seg_elem(const FinishArgs& a, int blk, bool& bad_g, bool& bad_w) {
  // 64-bit: the masked tensors TOGETHER may exceed 2^31 elements (configs[4]: W_ih 1.96e9 + W_hh 0.45e9)
  int64_t e = (int64_t)blk * 256 + threadIdx.x;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (!((a.elem_mask >> t) & 1)) continue;
    if (e < (int64_t)a.n[t]) {
      const int ncols = t == T_WIH ? a.I : 1;
      emit<ADAM>(a, t, e, (int)(e / ncols), (int)(e % ncols), a.g[t][e], false, bad_g, bad_w);
      return;
    }
    e -= a.n[t];
  }
}

template <int ADAM>   // 1: Adam + prepared images after the reduction; 0: reduce only
__global__ void __launch_bounds__(256) finish_kernel(const FinishArgs a) {
  int blk = blockIdx.x;
  bool bad_g = false, bad_w = false;
  if (blk < a.ih.nblocks) {
    if (a.ih.kind == 2) seg_tn<ADAM>(a, a.ih, T_WIH, T_BIH, blk, bad_g, bad_w);
    else seg_plain<ADAM>(a, a.ih, T_WIH, T_BIH, blk, bad_g, bad_w);
  } else if ((blk -= a.ih.nblocks) < a.hh.nblocks) {
    if (a.hh.kind == 2) seg_tn<ADAM>(a, a.hh, T_WHH, T_BHH, blk, bad_g, bad_w);
    else seg_plain<ADAM>(a, a.hh, T_WHH, T_BHH, blk, bad_g, bad_w);
  } else if ((blk -= a.hh.nblocks) < a.conv_blocks) {
    seg_conv<ADAM>(a, blk, bad_g, bad_w);
  } else {
    seg_elem<ADAM>(a, blk - a.conv_blocks, bad_g, bad_w);
  }
  report_status(a.status, bad_g, WGNN_STATUS_GRAD_NONFINITE);
  report_status(a.status, bad_w, WGNN_STATUS_WEIGHT_RANGE);
}

}  // namespace

int finish_seg_blocks(const FinSeg& s) {
  if (s.kind == 2) {                                                           // slab4 / 64, or slab4 / 256 (seg_tn, few K chunks)
    const size_t b64 = (size_t)s.ntiles * TN_WAVES * 5 * s.T;
    return (int)(s.splitk < 4 ? (b64 + 3) / 4 : b64);
  }
  if (s.kind == 1) return (int)(((size_t)s.Mgemm * s.pitch + 255) / 256);
  return 0;
}

// a.ih / a.hh with kind 0 and conv_partial == nullptr are skipped; elem_mask names the tensors read from g.
int launch_finish(FinishArgs a, hipStream_t st) {
  a.ih.nblocks = finish_seg_blocks(a.ih);
  a.hh.nblocks = finish_seg_blocks(a.hh);
  a.conv_blocks = a.conv_partial ? cdiv_i(PART, 32) : 0;
  int64_t ne = 0;
  for (int t = 0; t < 8; ++t)
    if ((a.elem_mask >> t) & 1) ne += a.n[t];
  const int64_t eb = (ne + 255) / 256;
  const int64_t grid64 = (int64_t)a.ih.nblocks + a.hh.nblocks + a.conv_blocks + eb;
  if (grid64 > 0x7fffffffll) return WGNN_ERR_SHAPE;                   // one launch's grid.x
  a.elem_blocks = (int)eb;
  const int grid = (int)grid64;
  if (grid < 1) return WGNN_OK;
  double by = 0.0;
  if (a.ih.kind) by += 4.0 * a.ih.splitk * a.ih.Mout * (double)a.ih.Nout;
  if (a.hh.kind) by += 4.0 * a.hh.splitk * a.hh.Mout * (double)a.hh.Nout;
  if (a.conv_partial) by += 4.0 * a.conv_rows * PART;
  double np = 0.0;
  for (int t = 0; t < 8; ++t) np += a.n[t];
  by += (a.adam ? 28.0 + (a.prep_kind ? 8.0 : 0.0) : 4.0) * np;
  if (a.adam)
    PROF_LAUNCH("finish_kernel<1>", 12.0 * np, by, st,
                hipLaunchKernelGGL(finish_kernel<1>, dim3(grid), dim3(256), 0, st, a));
  else
    PROF_LAUNCH("finish_kernel<0>", np, by, st,
                hipLaunchKernelGGL(finish_kernel<0>, dim3(grid), dim3(256), 0, st, a));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
