// General-shape kernels: everything the LDS/register-resident fast kernels do not cover.
//
//   * CSR adjacency (BASELINE config 5: 4096-station k-NN graph): the two GraphConvLayers
//     relu((A X) W + b)  (src/step5_gcn_layer_model.py:15,18,21) with A X as a row-gather SpMM, and
//     their backward, which needs A^T products and therefore the transposed CSR too.
//   * GRU recurrences of any hidden width (src/step6_gcn_gru_combined_model.py:23): per timestep one
//     fp32 MFMA GEMM (gemm.hip) for gh = W_hh h + b_hh / dh += dgh W_hh and one elementwise cell kernel.
//
// All arithmetic is fp32 in a fixed order (deterministic).  At these sizes the step is dominated by
// the W_ih / W_hh GEMMs (45 TFLOP per step per GPU at config 5 against 6 GFLOP of SpMM), so the SpMM
// kernels are written for clarity: one thread per (tile, station) row, 13 accumulators in registers,
// neighbour rows gathered from L2 (a tile's X is 213 KB at S = 4096).
#include "common.h"

namespace {

constexpr int F13 = 13;
constexpr int FP = 16;
constexpr int PART = 2 * FP * FP + 2 * FP;   // same partial layout as gcn.hip: dW1 | dW2 | db1 | db2
constexpr int ROWS = 256;                    // stations per block
constexpr int GEN_BLOCKS = 1024;             // persistent backward blocks = partial rows

struct Csr {
  const int* rowptr;
  const int* col;
  const float* val;
};

// p[f] = sum_e val[e] * V[col[e]][f] over row s, in CSR order
__device__ __forceinline__ void spmv_row(const Csr& c, int s, const float* __restrict__ V, float (&p)[F13]) {
#pragma unroll
  for (int f = 0; f < F13; ++f) p[f] = 0.f;
  const int e1 = c.rowptr[s + 1];
  for (int e = c.rowptr[s]; e < e1; ++e) {
    const float v = c.val[e];
    const float* x = V + (size_t)c.col[e] * F13;
#pragma unroll
    for (int f = 0; f < F13; ++f) p[f] = fmaf(v, x[f], p[f]);
  }
}

// Out[tile][s][:] = relu((A In[tile])[s][:] W + b).  PLANES: the output row is written as fp16 hi/lo planes
// [ntiles][ld_out] (the f16x3 interchange format, pgemm.hip) with 1.0 in column S*13 and zeros behind it.
template <bool PLANES>
__global__ void __launch_bounds__(ROWS) csr_layer_fwd_kernel(int ntiles, int S, Csr A, const float* __restrict__ In,
                                                             size_t ld_in, const float* __restrict__ W,
                                                             const float* __restrict__ b, float* __restrict__ Out,
                                                             size_t ld_out, _Float16* __restrict__ Ohi,
                                                             _Float16* __restrict__ Olo, unsigned* status) {
  bool bad = false;
  __shared__ float Ws[F13 * F13], bs[F13];
  for (int i = threadIdx.x; i < F13 * F13; i += ROWS) Ws[i] = W[i];
  if (threadIdx.x < F13) bs[threadIdx.x] = b[threadIdx.x];
  __syncthreads();
  const int s = blockIdx.x * ROWS + threadIdx.x;
  const int I = S * F13;
  for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
    if (PLANES && blockIdx.x == 0)                    // the ones column and the padding behind it
      for (int c = I + threadIdx.x; c < (int)ld_out; c += ROWS) {
        Ohi[(size_t)tile * ld_out + c] = (_Float16)(c == I ? 1.f : 0.f);
        if (Olo) Olo[(size_t)tile * ld_out + c] = (_Float16)0.f;
      }
    if (s >= S) continue;
    float p[F13];
    spmv_row(A, s, In + (size_t)tile * ld_in, p);
#pragma unroll
    for (int c = 0; c < F13; ++c) {
      float a = bs[c];
#pragma unroll
      for (int f = 0; f < F13; ++f) a = fmaf(p[f], Ws[f * F13 + c], a);
      a = fmaxf(a, 0.f);
      const size_t o = (size_t)tile * ld_out + (size_t)s * F13 + c;
      if (PLANES) {
        bad |= out_of_fp16_range(a);
        const _Float16 h = (_Float16)a;
        Ohi[o] = h;
        if (Olo) Olo[o] = (_Float16)(a - (float)h);
      } else {
        Out[o] = a;
      }
    }
  }
  if (PLANES) report_status(status, bad, WGNN_STATUS_ACT_RANGE);
}

// One backward layer for a block's share of (tile, row-block) items:
//   dz = dOut * (Out > 0)           where dOut is given directly (layer 2: dg) or as A^T DU (layer 1)
//   p  = (A In)[s]                  (recomputed aggregation)
//   dW += p^T dz, db += dz          (per-block partials, fixed order)
//   DU[tile][s][:] = dz W^T         (layer 2 only: what layer 1's A^T product consumes)
template <bool LAYER2>
__global__ void __launch_bounds__(ROWS) csr_layer_bwd_kernel(int ntiles, int S, Csr A, Csr AT,
                                                             const float* __restrict__ In, size_t ld_in,
                                                             const float* __restrict__ Out,
                                                             const _Float16* __restrict__ Out_hi, size_t ld_out,
                                                             const float* __restrict__ dOut, size_t ld_dout,
                                                             const float* __restrict__ W, float* __restrict__ DU,
                                                             const float* __restrict__ scales,
                                                             float* __restrict__ partial) {
  __shared__ float Ws[F13 * F13];
  __shared__ float ps[ROWS][F13], dzs[ROWS][F13];
  const int tid = threadIdx.x;
  if (LAYER2)
    for (int i = tid; i < F13 * F13; i += ROWS) Ws[i] = W[i];
  const int nrb = (S + ROWS - 1) / ROWS;
  const long long nitems = (long long)ntiles * nrb;
  // thread t < 169: dW[t/13][t%13]; 169 <= t < 182: db[t-169].  fp64: these are long sums of mixed-sign terms
  // (cancellation ~100x at S = 4096) and the loop is far off the critical path.
  double acc = 0.0;
  for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int tile = (int)(item / nrb), s = (int)(item % nrb) * ROWS + tid;
    float p[F13], dz[F13];
    if (s < S) {
      const size_t oo = (size_t)tile * ld_out + (size_t)s * F13;
      bool on[F13];                                           // ReLU mask: fp32 activations or their fp16 hi plane
#pragma unroll
      for (int f = 0; f < F13; ++f) on[f] = Out_hi ? (float)Out_hi[oo + f] > 0.f : Out[oo + f] > 0.f;
      if (LAYER2) {
        const float* d = dOut + (size_t)tile * ld_dout + (size_t)s * F13;
#pragma unroll
        for (int f = 0; f < F13; ++f) dz[f] = on[f] ? d[f] : 0.f;
      } else {
        spmv_row(AT, s, dOut + (size_t)tile * ld_dout, dz);   // dH1 = A^T DU
#pragma unroll
        for (int f = 0; f < F13; ++f) dz[f] = on[f] ? dz[f] : 0.f;
      }
      spmv_row(A, s, In + (size_t)tile * ld_in, p);
    } else {
#pragma unroll
      for (int f = 0; f < F13; ++f) p[f] = dz[f] = 0.f;
    }
    __syncthreads();   // previous item's outer product is done (and Ws is loaded)
#pragma unroll
    for (int f = 0; f < F13; ++f) {
      ps[tid][f] = p[f];
      dzs[tid][f] = dz[f];
    }
    if (LAYER2 && s < S) {
      float* du = DU + ((size_t)tile * S + s) * F13;
#pragma unroll
      for (int f = 0; f < F13; ++f) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < F13; ++c) a = fmaf(dz[c], Ws[f * F13 + c], a);
        du[f] = a;
      }
    }
    __syncthreads();
    if (tid < F13 * F13) {
      const int f = tid / F13, c = tid % F13;
      for (int r = 0; r < ROWS; ++r) acc += (double)ps[r][f] * (double)dzs[r][c];
    } else if (tid < F13 * F13 + F13) {
      const int c = tid - F13 * F13;
      for (int r = 0; r < ROWS; ++r) acc += (double)dzs[r][c];
    }
  }
  if (scales) acc *= (double)scales[1];   // dOut arrives in the backward's power-of-two scaled units
  float* mine = partial + (size_t)blockIdx.x * PART;
  if (tid < F13 * F13) mine[(LAYER2 ? FP * FP : 0) + (tid / F13) * FP + tid % F13] = (float)acc;
  else if (tid < F13 * F13 + F13) mine[2 * FP * FP + (LAYER2 ? FP : 0) + tid - F13 * F13] = (float)acc;
}

// ---- GRU cell, forward: one thread per (window, hidden unit) -----------------------------------
__global__ void gru_cell_fwd_kernel(int B, int T, int t, int H, const float* __restrict__ GI, int ldgi,
                                    const float* __restrict__ GH, int ldgh, const float* __restrict__ bhh,
                                    float* __restrict__ Y, float* __restrict__ gates) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H) return;
  const int b = (int)(i / H), j = (int)(i % H);
  const size_t bt = (size_t)b * T + t;
  const float* gi = GI + bt * ldgi;
  float ghr, ghz, ghn, hprev = 0.f;
  if (t == 0) {   // h0 = 0: gh = b_hh
    ghr = bhh[j]; ghz = bhh[H + j]; ghn = bhh[2 * H + j];
  } else {
    const float* gh = GH + (size_t)b * ldgh;
    ghr = gh[j]; ghz = gh[H + j]; ghn = gh[2 * H + j];
    hprev = Y[(bt - 1) * H + j];
  }
  const float r = sigmoidf_(gi[j] + ghr);
  const float z = sigmoidf_(gi[H + j] + ghz);
  const float n = tanhf_(gi[2 * H + j] + r * ghn);
  Y[bt * H + j] = (1.f - z) * n + z * hprev;
  if (gates) {
    float* gp = gates + bt * 4 * H;
    gp[j] = r; gp[H + j] = z; gp[2 * H + j] = n; gp[3 * H + j] = ghn;
  }
}

// backward cell: dh = dY_t + dhz + dhw (the two carries of step t+1)
__global__ void gru_cell_bwd_kernel(int B, int T, int t, int H, const float* __restrict__ Y,
                                    const float* __restrict__ dY, const float* __restrict__ gates,
                                    float* __restrict__ dhz, const float* __restrict__ dhw, float* __restrict__ dGI,
                                    float* __restrict__ dGH, int ldd) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H) return;
  const int b = (int)(i / H), j = (int)(i % H);
  const size_t bt = (size_t)b * T + t;
  float dh = dY[bt * H + j];
  if (t < T - 1) dh += dhz[i] + (dhw ? dhw[i] : 0.f);
  const float* gp = gates + bt * 4 * H;
  const float r = gp[j], z = gp[H + j], n = gp[2 * H + j], ghn = gp[3 * H + j];
  const float hprev = t > 0 ? Y[(bt - 1) * H + j] : 0.f;
  const float dn = dh * (1.f - z);
  const float dzg = dh * (hprev - n);
  const float dnt = dn * (1.f - n * n);
  const float dr = dnt * ghn;
  const float dar = dr * r * (1.f - r);
  const float daz = dzg * z * (1.f - z);
  float* gi = dGI + bt * ldd;
  float* gh = dGH + bt * ldd;
  gi[j] = dar; gi[H + j] = daz; gi[2 * H + j] = dnt;
  gh[j] = dar; gh[H + j] = daz; gh[2 * H + j] = dnt * r;
  dhz[i] = dh * z;
}

// ---- f16x3 variants of the cells: h_t and dgi / dgh are (also) written as fp16 hi/lo planes, the operand
// format of the plane GEMMs (pgemm.hip).  Thread (b, j) with j over the padded plane width.
__global__ void gru_cell_fwd_x3_kernel(int B, int T, int t, int H, int Hp, const float* __restrict__ GI, int ldgi,
                                       const float* __restrict__ GH, int ldgh, const float* __restrict__ bhh,
                                       float* __restrict__ Y, float* __restrict__ gates, _Float16* __restrict__ yhi,
                                       _Float16* __restrict__ ylo, _Float16* __restrict__ chi,
                                       _Float16* __restrict__ clo) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * Hp) return;
  const int b = (int)(i / Hp), j = (int)(i % Hp);
  const size_t bt = (size_t)b * T + t;
  if (t == 0 && b == 0) {          // the extra plane row B*T: what [Hprev | 1] is at a window start
    yhi[(size_t)B * T * Hp + j] = (_Float16)(j == H ? 1.f : 0.f);
    if (ylo) ylo[(size_t)B * T * Hp + j] = (_Float16)0.f;
  }
  if (j >= H) {                    // ones column (-> b_hh / db_hh in the GEMMs) and zero padding
    yhi[bt * Hp + j] = chi[i] = (_Float16)(j == H ? 1.f : 0.f);
    if (ylo) ylo[bt * Hp + j] = clo[i] = (_Float16)0.f;
    return;
  }
  const float* gi = GI + bt * ldgi;
  float ghr, ghz, ghn, hprev = 0.f;
  if (t == 0) {
    ghr = bhh[j]; ghz = bhh[H + j]; ghn = bhh[2 * H + j];
  } else {
    const float* gh = GH + (size_t)b * ldgh;
    ghr = gh[j]; ghz = gh[H + j]; ghn = gh[2 * H + j];
    hprev = Y[(bt - 1) * H + j];
  }
  const float r = sigmoid_fast(gi[j] + ghr);
  const float z = sigmoid_fast(gi[H + j] + ghz);
  const float n = tanh_fast(gi[2 * H + j] + r * ghn);
  const float h = (1.f - z) * n + z * hprev;
  Y[bt * H + j] = h;
  const _Float16 hh = (_Float16)h;
  yhi[bt * Hp + j] = chi[i] = hh;       // c*: the same row in a compact [B][Hp] image, the next step's GEMM operand
  if (ylo) ylo[bt * Hp + j] = clo[i] = (_Float16)(h - (float)hh);
  if (gates) {
    float* gp = gates + bt * 4 * H;
    gp[j] = r; gp[H + j] = z; gp[2 * H + j] = n; gp[3 * H + j] = ghn;
  }
}

__device__ __forceinline__ void put_planes(_Float16* hi, _Float16* lo, size_t idx, float v) {
  const _Float16 h = (_Float16)v;
  hi[idx] = h;
  if (lo) lo[idx] = (_Float16)(v - (float)h);
}

// dY is scaled by scales[0] on the way in; everything downstream stays in those units
__global__ void gru_cell_bwd_x3_kernel(int B, int T, int t, int H, const float* __restrict__ Y,
                                       const float* __restrict__ dY, const float* __restrict__ gates,
                                       const float* __restrict__ scales, float* __restrict__ dhz,
                                       const float* __restrict__ dhw, _Float16* __restrict__ gihi,
                                       _Float16* __restrict__ gilo, _Float16* __restrict__ ghhi,
                                       _Float16* __restrict__ ghlo, int ldd, _Float16* __restrict__ chi,
                                       _Float16* __restrict__ clo) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H) return;
  const int b = (int)(i / H), j = (int)(i % H);
  const size_t bt = (size_t)b * T + t;
  float dh = dY[bt * H + j] * scales[0];
  if (t < T - 1) dh += dhz[i] + dhw[i];
  const float* gp = gates + bt * 4 * H;
  const float r = gp[j], z = gp[H + j], n = gp[2 * H + j], ghn = gp[3 * H + j];
  const float hprev = t > 0 ? Y[(bt - 1) * H + j] : 0.f;
  const float dn = dh * (1.f - z);
  const float dzg = dh * (hprev - n);
  const float dnt = dn * (1.f - n * n);
  const float dr = dnt * ghn;
  const float dar = dr * r * (1.f - r);
  const float daz = dzg * z * (1.f - z);
  const size_t row = bt * ldd;
  put_planes(gihi, gilo, row + j, dar);
  put_planes(gihi, gilo, row + H + j, daz);
  put_planes(gihi, gilo, row + 2 * H + j, dnt);
  put_planes(ghhi, ghlo, row + j, dar);
  put_planes(ghhi, ghlo, row + H + j, daz);
  put_planes(ghhi, ghlo, row + 2 * H + j, dnt * r);
  const size_t crow = (size_t)b * ldd;         // compact [B][ldd] image of this step's dgh: the GEMM operand
  put_planes(chi, clo, crow + j, dar);
  put_planes(chi, clo, crow + H + j, daz);
  put_planes(chi, clo, crow + 2 * H + j, dnt * r);
  for (int c = 3 * H + j; c < ldd; c += H) {   // K padding of the plane rows
    put_planes(gihi, gilo, row + c, 0.f);
    put_planes(ghhi, ghlo, row + c, 0.f);
    put_planes(chi, clo, crow + c, 0.f);
  }
  dhz[i] = dh * z;
}

// Out[tile][s][:] = (M In[tile])[s][:]   (the layer's dX = A^T (dZ W^T) when GraphConvLayer is used on its own)
__global__ void __launch_bounds__(ROWS) csr_spmm_kernel(int ntiles, int S, Csr M, const float* __restrict__ In,
                                                        float* __restrict__ Out) {
  const int s = blockIdx.x * ROWS + threadIdx.x;
  if (s >= S) return;
  const size_t I = (size_t)S * F13;
  for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
    float p[F13];
    spmv_row(M, s, In + (size_t)tile * I, p);
    float* o = Out + (size_t)tile * I + (size_t)s * F13;
#pragma unroll
    for (int f = 0; f < F13; ++f) o[f] = p[f];
  }
}

Csr csr_of(const void* blob, int S, int nnz, bool transposed) {
  const int* w = (const int*)blob + (transposed ? (size_t)S + 1 + 2 * (size_t)nnz : 0);
  Csr c;
  c.rowptr = w;
  c.col = w + S + 1;
  c.val = (const float*)(w + S + 1 + nnz);
  return c;
}

}  // namespace

size_t gcn_csr_bwd_partial_floats() { return (size_t)GEN_BLOCKS * PART; }

// g: fp32 [ntiles][ldg], or (g_planes != nullptr) fp16 hi/lo planes [ntiles][ldg] each (lo skipped if !x3)
int launch_gcn2_csr_fwd(int ntiles, int S, int nnz, const void* csr, const float* X, const float* W1, const float* b1,
                        const float* W2, const float* b2, float* h1, float* g, void* g_planes, size_t ldg, bool x3,
                        unsigned* status, hipStream_t st) {
  const Csr A = csr_of(csr, S, nnz, false);
  const dim3 grid(cdiv_i(S, ROWS), ntiles < 16384 ? ntiles : 16384);
  const double fl = (double)ntiles * (2.0 * nnz * F13 + 2.0 * S * F13 * F13), by = (double)ntiles * S * F13 * 8.0;
  const size_t I = (size_t)S * F13;
  PROF_LAUNCH("csr_layer_fwd_kernel", fl, by, st,
              hipLaunchKernelGGL(csr_layer_fwd_kernel<false>, grid, dim3(ROWS), 0, st, ntiles, S, A, X, I, W1, b1, h1, I,
                                 (_Float16*)nullptr, (_Float16*)nullptr, (unsigned*)nullptr));
  WGNN_CHECK_LAUNCH();
  if (g_planes) {
    _Float16* hi = (_Float16*)g_planes;
    PROF_LAUNCH("csr_layer_fwd_kernel", fl, by, st,
                hipLaunchKernelGGL(csr_layer_fwd_kernel<true>, grid, dim3(ROWS), 0, st, ntiles, S, A, h1, I, W2, b2,
                                   (float*)nullptr, ldg, hi, x3 ? hi + (size_t)ntiles * ldg : (_Float16*)nullptr, status));
  } else {
    PROF_LAUNCH("csr_layer_fwd_kernel", fl, by, st,
                hipLaunchKernelGGL(csr_layer_fwd_kernel<false>, grid, dim3(ROWS), 0, st, ntiles, S, A, h1, I, W2, b2, g,
                                   ldg, (_Float16*)nullptr, (_Float16*)nullptr, (unsigned*)nullptr));
  }
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

// dW1, db1, dW2, db2 of the two layers given dg; du: scratch [ntiles][S*13]; partial: gcn_csr_bwd_partial_floats().
// g is fp32 [ntiles][ldg] or (g_hi != nullptr) the fp16 hi plane; scales != nullptr: dg is in scaled units.
int launch_gcn2_csr_bwd(int ntiles, int S, int nnz, const void* csr, const float* X, const float* W2, const float* h1,
                        const float* g, const void* g_hi, size_t ldg, const float* dg, size_t ld_dg,
                        const float* scales, float* du, float* partial, float* dW1, float* db1, float* dW2, float* db2,
                        hipStream_t st) {
  const Csr A = csr_of(csr, S, nnz, false), AT = csr_of(csr, S, nnz, true);
  const size_t I = (size_t)S * F13;
  const double fl = (double)ntiles * (2.0 * nnz * F13 + 4.0 * S * F13 * F13), by = (double)ntiles * S * F13 * 16.0;
  PROF_LAUNCH("csr_layer_bwd_kernel<2>", fl, by, st,
              hipLaunchKernelGGL((csr_layer_bwd_kernel<true>), dim3(GEN_BLOCKS), dim3(ROWS), 0, st, ntiles, S, A, AT, h1,
                                 I, g, (const _Float16*)g_hi, ldg, dg, ld_dg, W2, du, scales, partial));
  WGNN_CHECK_LAUNCH();
  PROF_LAUNCH("csr_layer_bwd_kernel<1>", fl + (double)ntiles * 2.0 * nnz * F13, by, st,
              hipLaunchKernelGGL((csr_layer_bwd_kernel<false>), dim3(GEN_BLOCKS), dim3(ROWS), 0, st, ntiles, S, A, AT, X,
                                 I, h1, (const _Float16*)nullptr, I, du, I, (const float*)nullptr, (float*)nullptr,
                                 scales, partial));
  WGNN_CHECK_LAUNCH();
  if (!dW1) return WGNN_OK;              // deferred: finish.hip reduces the gcn_csr_bwd_rows() partial rows
  return launch_gcn_partial_reduce(partial, GEN_BLOCKS, dW1, db1, dW2, db2, nullptr, st);
}

int gcn_csr_bwd_rows() { return GEN_BLOCKS; }

// Y, gates from GI: per step gh = Hprev W_hh^T + b_hh (GEMM, skipped at t = 0 where h = 0) and the cell.
int launch_gru_gen_fwd(int B, int T, int H, const float* GI, int ldgi, const float* Whh, const float* bhh, float* Y,
                       float* gates, float* gh /*[B][3H]*/, hipStream_t st) {
  const int nb = (int)(((long long)B * H + 255) / 256);
  for (int t = 0; t < T; ++t) {
    if (t > 0) {
      GemmArgs a = {};
      a.A = Y + (size_t)(t - 1) * H; a.lda = T * H; a.a_kcontig = 1;     // row b = h_{t-1} of window b
      a.B = Whh; a.ldb = H; a.b_kcontig = 1;
      a.C = gh; a.ldc = 3 * H; a.M = B; a.N = 3 * H; a.K = H;
      a.bias = bhh; a.splitk = 1;
      int rc = launch_gemm_f32(a, st);
      if (rc != WGNN_OK) return rc;
    }
    PROF_LAUNCH("gru_cell_fwd_kernel", 0.0, (double)B * H * 4.0 * 9, st,
                hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(nb), dim3(256), 0, st, B, T, t, H, GI, ldgi, gh, 3 * H, bhh,
                                   Y, gates));
    WGNN_CHECK_LAUNCH();
  }
  return WGNN_OK;
}

// dGI, dGH rows for all (b, t); dhz / dhw: scratch [B][H] each
int launch_gru_gen_bwd(int B, int T, int H, const float* Whh, const float* Y, const float* dY, const float* gates,
                       float* dGI, float* dGH, int ldd, float* dhz, float* dhw, hipStream_t st) {
  const int nb = (int)(((long long)B * H + 255) / 256);
  for (int t = T - 1; t >= 0; --t) {
    PROF_LAUNCH("gru_cell_bwd_kernel", 0.0, (double)B * H * 4.0 * 14, st,
                hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(nb), dim3(256), 0, st, B, T, t, H, Y, dY, gates, dhz, dhw,
                                   dGI, dGH, ldd));
    WGNN_CHECK_LAUNCH();
    if (t > 0) {   // dhw = dGH_t W_hh: the recurrent part of dh_{t-1}
      GemmArgs a = {};
      a.A = dGH + (size_t)t * ldd; a.lda = T * ldd; a.a_kcontig = 1;
      a.B = Whh; a.ldb = H; a.b_kcontig = 0;
      a.C = dhw; a.ldc = H; a.M = B; a.N = H; a.K = 3 * H; a.splitk = 1;
      int rc = launch_gemm_f32(a, st);
      if (rc != WGNN_OK) return rc;
    }
  }
  return WGNN_OK;
}

// ---- f16x3: the per-step products run on the plane GEMMs -----------------------------------------
// whh_planes: split(W_hh | b_hh) as launch_split_weight2(w_hh, 3H, H, 0, b_hh, H, ., pgemm_nt_np(3H), Hp) makes it;
// y_planes: 2 x [B*T+1][Hp] halfs (hi, lo); gh: [B][ldgi] floats.
int launch_gru_gen_fwd_x3(int B, int T, int H, const float* GI, int ldgi, const void* whh_planes, int np_g3,
                          const float* bhh, float* Y, float* gates, void* y_planes, float* gh, float* kpart,
                          void* hc /*B*Hp floats*/, bool x3, hipStream_t st) {
  const int Hp = grux_hp(H);
  _Float16* yhi = (_Float16*)y_planes;
  _Float16* ylo = yhi + ((size_t)B * T + 1) * Hp;
  // The GEMM reads h_{t-1} from a compact [B][Hp] copy: rows of the [B*T][Hp] planes are T*Hp apart, and 64-byte
  // row segments that far apart (0.6 MB at H = 12288) cost 5x in the staging loads.
  _Float16* chi = (_Float16*)hc;
  _Float16* clo = chi + (size_t)B * Hp;
  const int nb = (int)(((long long)B * Hp + 255) / 256);
  for (int t = 0; t < T; ++t) {
    if (t > 0) {   // gh = [h_{t-1} | 1] (W_hh | b_hh)^T
      int rc = launch_pgemm_nt(chi, clo, Hp, B, Hp, whh_planes, np_g3, gh, ldgi, 3 * H, nullptr, x3, kpart, st);
      if (rc != WGNN_OK) return rc;
    }
    PROF_LAUNCH("gru_cell_fwd_x3_kernel", 0.0, (double)B * H * 4.0 * 10, st,
                hipLaunchKernelGGL(gru_cell_fwd_x3_kernel, dim3(nb), dim3(256), 0, st, B, T, t, H, Hp, GI, ldgi, gh, ldgi,
                                   bhh, Y, gates, yhi, x3 ? ylo : (_Float16*)nullptr, chi, clo));
    WGNN_CHECK_LAUNCH();
  }
  return WGNN_OK;
}

// whhT_planes: split(W_hh^T) as launch_split_weight2(w_hh, 3H, H, 1, nullptr, 0, ., pgemm_nt_np(H), ldd) makes it;
// dgi / dgh planes: hi [B*T][ldd] followed by lo [B*T][ldd]
int launch_gru_gen_bwd_x3(int B, int T, int H, const void* whhT_planes, int np_h, const float* Y, const float* dY,
                          const float* gates, const float* scales, void* dgi_planes, void* dgh_planes, int ldd,
                          float* dhz, float* dhw, float* kpart, void* dc /*B*ldd floats*/, bool x3, hipStream_t st) {
  const size_t PG = (size_t)B * T * ldd;
  _Float16* chi = (_Float16*)dc;               // compact [B][ldd] planes of the current step's dgh
  _Float16* clo = chi + (size_t)B * ldd;
  _Float16* gihi = (_Float16*)dgi_planes;
  _Float16* ghhi = (_Float16*)dgh_planes;
  _Float16* gilo = x3 ? gihi + PG : nullptr;
  _Float16* ghlo = x3 ? ghhi + PG : nullptr;
  const int nb = (int)(((long long)B * H + 255) / 256);
  for (int t = T - 1; t >= 0; --t) {
    PROF_LAUNCH("gru_cell_bwd_x3_kernel", 0.0, (double)B * H * 4.0 * 14, st,
                hipLaunchKernelGGL(gru_cell_bwd_x3_kernel, dim3(nb), dim3(256), 0, st, B, T, t, H, Y, dY, gates, scales,
                                   dhz, dhw, gihi, gilo, ghhi, ghlo, ldd, chi, x3 ? clo : (_Float16*)nullptr));
    WGNN_CHECK_LAUNCH();
    if (t > 0) {   // dhw = dGH_t W_hh, stays in scaled units
      int rc = launch_pgemm_nt(chi, clo, ldd, B, ldd, whhT_planes, np_h, dhw, H, H, nullptr, x3, kpart, st);
      if (rc != WGNN_OK) return rc;
    }
  }
  return WGNN_OK;
}

// ---- one GraphConvLayer with a CSR adjacency (wgnn_gcn_layer_csr_fwd / _bwd) ------------------------
size_t gcn1_csr_bwd_ws_floats(int ntiles, int S) { return (size_t)ntiles * S * F13 + gcn_csr_bwd_partial_floats(); }

int launch_gcn1_csr_fwd(int ntiles, int S, int nnz, const void* csr, const float* X, const float* W, const float* b,
                        float* out, hipStream_t st) {
  const Csr A = csr_of(csr, S, nnz, false);
  const dim3 grid(cdiv_i(S, ROWS), ntiles < 16384 ? ntiles : 16384);
  const size_t I = (size_t)S * F13;
  hipLaunchKernelGGL(csr_layer_fwd_kernel<false>, grid, dim3(ROWS), 0, st, ntiles, S, A, X, I, W, b, out, I,
                     (_Float16*)nullptr, (_Float16*)nullptr, (unsigned*)nullptr);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

// dW, db (overwritten) and, if dX != nullptr, dX = A^T ((dout * (out > 0)) W^T); ws: gcn1_csr_bwd_ws_floats()
int launch_gcn1_csr_bwd(int ntiles, int S, int nnz, const void* csr, const float* X, const float* W, const float* out,
                        const float* dout, float* dW, float* db, float* dX, float* ws, hipStream_t st) {
  const Csr A = csr_of(csr, S, nnz, false), AT = csr_of(csr, S, nnz, true);
  const size_t I = (size_t)S * F13;
  float* du = ws;
  float* partial = ws + (size_t)ntiles * I;
  hipLaunchKernelGGL((csr_layer_bwd_kernel<true>), dim3(GEN_BLOCKS), dim3(ROWS), 0, st, ntiles, S, A, AT, X, I, out,
                     (const _Float16*)nullptr, I, dout, I, W, du, (const float*)nullptr, partial);
  WGNN_CHECK_LAUNCH();
  int rc = launch_gcn_partial_reduce(partial, GEN_BLOCKS, nullptr, nullptr, dW, db, nullptr, st);   // the "layer 2" slots
  if (rc != WGNN_OK || !dX) return rc;
  const dim3 grid(cdiv_i(S, ROWS), ntiles < 16384 ? ntiles : 16384);
  hipLaunchKernelGGL(csr_spmm_kernel, grid, dim3(ROWS), 0, st, ntiles, S, AT, du, dX);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
