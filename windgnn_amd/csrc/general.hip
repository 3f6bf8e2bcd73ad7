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

struct Csr {
  const int* rowptr;
  const int* col;
  const float* val;
};

// p[f] = sum_e val[e] * V[col[e]][f] over row s, in CSR order (csr_spmm_kernel: the one-layer API's dX)
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

// ---- LDS-tiled SpMM layers --------------------------------------------------------------------------------------------
// One 1024-thread block per CU.  A block item is a set of up to RPT * 1024 output rows that share one gather source: a row
// block of one tile (S >= RPT * 1024) or the rows of floor(RPT * 1024 / S) consecutive tiles (smaller graphs; their [S][13]
// inputs are contiguous in memory, so the group is one [ng * S][13] matrix and tile g's columns are offset by g * S).  The
// source is staged through LDS in chunks of CH = 2048 rows padded to 16 floats (128 KB; coalesced 16-byte global loads),
// and every thread walks the CSR rows it owns once per chunk, gathering the neighbours that fall into the chunk with four
// ds_read_b128 each: the scattered 52-byte reads that bound the thread-per-row kernel of rounds 1-4 on the texture path (64
// cache lines per wave instruction: 2.0 ms per layer at S = 4096, 128 windows) hit LDS banks instead.  With S <= CH the sum
// runs in CSR order; beyond that in chunk order (CSR order inside a chunk) -- fixed either way.
constexpr int CT = 1024;                  // threads per block
constexpr int CH = 2048;                  // source rows per LDS chunk
constexpr int CSR_BLOCKS = 256;           // persistent blocks (= partial rows of the backward)
constexpr size_t CSR_LDS = ((size_t)CH * 16 + 256) * sizeof(float);       // chunk | W (169) | b (13)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct CsrItem {
  int tile0, ng, row0, nrows, VC;         // first tile, tiles in the group, first row, output rows, source rows
};
// item -> rows; RB = rows per item
__device__ __forceinline__ CsrItem csr_item(int item, int ntiles, int S, int RB) {
  CsrItem it;
  if (S >= RB) {
    const int nrb = (S + RB - 1) / RB;
    it.tile0 = item / nrb;
    it.ng = 1;
    it.row0 = (item - it.tile0 * nrb) * RB;
    it.nrows = min(RB, S - it.row0);
    it.VC = S;
  } else {
    const int G = RB / S;
    it.tile0 = item * G;
    it.ng = min(G, ntiles - it.tile0);
    it.row0 = 0;
    it.nrows = it.ng * S;
    it.VC = it.nrows;
  }
  return it;
}
// items of a launch (host and device); ntiles * row blocks stays far below 2^31 (api.hip bounds B * T)
__host__ __device__ inline int csr_items(int ntiles, int S, int RB) {
  return S >= RB ? ntiles * ((S + RB - 1) / RB) : (ntiles + RB / S - 1) / (RB / S);
}

// p[q][:] = sum over the CSR entries [e0[q], e1[q]) of val * src[col + coff[q]][:], src = [VC][13] contiguous.
template <int RPT, int NB = 4>
__device__ __forceinline__ void csr_gather(const Csr& c, const float* __restrict__ src, int VC, const int (&e0)[RPT],
                                           const int (&e1)[RPT], const int (&coff)[RPT], float* tile,
                                           float (&p)[RPT][F13]) {
#pragma unroll
  for (int q = 0; q < RPT; ++q)
#pragma unroll
    for (int f = 0; f < F13; ++f) p[q][f] = 0.f;
  for (int c0 = 0; c0 < VC; c0 += CH) {
    const int nr = min(CH, VC - c0), nf = nr * F13;
    const float* s0 = src + (size_t)c0 * F13;
    __syncthreads();                                   // whatever the buffer held has been consumed
    for (int j = 4 * (int)threadIdx.x; j < nf; j += 4 * CT) {
      if (j + 3 < nf) {
        const f32x4u v = *(const f32x4u*)(s0 + j);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned jj = (unsigned)(j + k), row = jj / F13;
          tile[row * 16 + (jj - row * F13)] = v[k];
        }
      } else {
        for (int k = 0; j + k < nf; ++k) {
          const unsigned jj = (unsigned)(j + k), row = jj / F13;
          tile[row * 16 + (jj - row * F13)] = s0[jj];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      for (int e = e0[q]; e < e1[q]; e += NB) {        // NB column indices per round trip
        int cc[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) cc[k] = e + k < e1[q] ? c.col[e + k] + coff[q] - c0 : -1;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if ((unsigned)cc[k] < (unsigned)nr) {
            const float v = c.val[e + k];
            const f32x4* x = (const f32x4*)(tile + cc[k] * 16);
            const f32x4 x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              p[q][f] = fmaf(v, x0[f], p[q][f]);
              p[q][4 + f] = fmaf(v, x1[f], p[q][4 + f]);
              p[q][8 + f] = fmaf(v, x2[f], p[q][8 + f]);
            }
            p[q][12] = fmaf(v, x3[0], p[q][12]);
          }
        }
      }
    }
  }
}

// The q-th row a thread owns is row v = tid + CT q of the item: tile it.tile0 + g, CSR row s (recomputed where needed: a
// division is cheaper than two live registers per row next to the 13-wide accumulators)
__device__ __forceinline__ bool csr_row_of(const CsrItem& it, int S, int q, int& g, int& srow) {
  const int v = (int)threadIdx.x + CT * q;
  g = it.ng == 1 ? 0 : v / S;
  srow = it.row0 + v - g * S;
  return v < it.nrows;
}
template <int RPT>
__device__ __forceinline__ void csr_rows(const Csr& c, const CsrItem& it, int S, int (&e0)[RPT], int (&e1)[RPT],
                                         int (&coff)[RPT]) {
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    int g, srow;
    const bool ok = csr_row_of(it, S, q, g, srow);
    const int sc = ok ? srow : 0;                              // rows past the item: an empty range (unconditional loads)
    e0[q] = c.rowptr[sc];
    e1[q] = ok ? c.rowptr[sc + 1] : e0[q];
    coff[q] = ok ? g * S : 0;
  }
}

// Out[tile][s][:] = relu((A In[tile])[s][:] W + b); In = [ntiles][S*13].  PLANES: the output row is written as fp16 hi/lo
// planes [ntiles][ld_out] (the f16x3 interchange format, pgemm.hip) with 1.0 in column S*13 and zeros behind it.  The rows go
// through LDS once more on their way out, so that the stores are whole 16-byte (planes: 8-byte) pieces of consecutive memory.
template <bool PLANES>
__global__ void __launch_bounds__(CT) csr_layer_fwd_kernel(int ntiles, int S, Csr A, const float* __restrict__ In,
                                                           const float* __restrict__ W, const float* __restrict__ b,
                                                           float* __restrict__ Out, size_t ld_out,
                                                           _Float16* __restrict__ Ohi, _Float16* __restrict__ Olo,
                                                           unsigned* status) {
#ifndef CSR_FWD_RPT
#define CSR_FWD_RPT 2
#endif
#ifndef CSR_FWD_NB
#define CSR_FWD_NB 4
#endif
  constexpr int RPT = CSR_FWD_RPT, RB = CT * RPT;              // output rows per item
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile = lds;
  float* Ws = lds + CH * 16;
  float* bs = Ws + 176;
  const int tid = threadIdx.x;
  for (int i = tid; i < F13 * F13; i += CT) Ws[i] = W[i];
  if (tid < F13) bs[tid] = b[tid];
  bool bad = false;
  const int I = S * F13;
  const bool vec = (I & 3) == 0 && (ld_out & 3) == 0;
  const int nitems = csr_items(ntiles, S, RB);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const CsrItem it = csr_item(item, ntiles, S, RB);
    float p[RPT][F13];
    {
      int e0[RPT], e1[RPT], coff[RPT];
      csr_rows<RPT>(A, it, S, e0, e1, coff);
      csr_gather<RPT, CSR_FWD_NB>(A, In + (size_t)it.tile0 * I, it.VC, e0, e1, coff, tile, p);
    }
#pragma unroll
    for (int h = 0; h < (RPT + 1) / 2; ++h) {                   // 2 CT rows at a time go out through LDS as [row][13]
      if (h * 2 * CT >= it.nrows) break;                       // block-uniform
      __syncthreads();                                         // the last chunk / the previous rows have been read
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = 2 * h + qq;
        if (q >= RPT || tid + CT * q >= it.nrows) continue;
        float* o = tile + (size_t)(tid + CT * qq) * F13;
#pragma unroll
        for (int c = 0; c < F13; ++c) {
          float a = bs[c];
#pragma unroll
          for (int f = 0; f < F13; ++f) a = fmaf(p[q < RPT ? q : 0][f], Ws[f * F13 + c], a);
          o[c] = fmaxf(a, 0.f);
        }
      }
      __syncthreads();
      const int lo = h * 2 * CT * F13, hi = min(it.nrows, (h + 1) * 2 * CT) * F13;   // linear elements of the item's output
      auto dst_of = [&](int j) -> size_t {                     // element j of the item -> offset in Out / the planes
        if (it.ng == 1) return (size_t)it.tile0 * ld_out + (size_t)it.row0 * F13 + j;
        const int g = j / I;
        return (size_t)(it.tile0 + g) * ld_out + (j - g * I);
      };
      if (vec) {
        for (int j = lo + 4 * tid; j < hi; j += 4 * CT) {
          const f32x4 v = *(const f32x4*)(tile + (j - lo));
          const size_t d = dst_of(j);
          if (PLANES) {
            f16x4 vh, vl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              bad |= out_of_fp16_range(v[k]);
              vh[k] = (_Float16)v[k];
              vl[k] = (_Float16)(v[k] - (float)vh[k]);
            }
            *(f16x4*)(Ohi + d) = vh;
            if (Olo) *(f16x4*)(Olo + d) = vl;
          } else {
            *(f32x4*)(Out + d) = v;
          }
        }
      } else {
        for (int j = lo + tid; j < hi; j += CT) {
          const float a = tile[j - lo];
          const size_t d = dst_of(j);
          if (PLANES) {
            bad |= out_of_fp16_range(a);
            const _Float16 hh = (_Float16)a;
            Ohi[d] = hh;
            if (Olo) Olo[d] = (_Float16)(a - (float)hh);
          } else {
            Out[d] = a;
          }
        }
      }
    }
    if (PLANES && it.row0 == 0) {                              // the ones column and the padding behind it
      const int padw = (int)ld_out - I;
      for (int j = tid; j < it.ng * padw; j += CT) {
        const int g = j / padw, c = I + j - g * padw;
        const size_t d = (size_t)(it.tile0 + g) * ld_out + c;
        Ohi[d] = (_Float16)(c == I ? 1.f : 0.f);
        if (Olo) Olo[d] = (_Float16)0.f;
      }
    }
  }
  if (PLANES) report_status(status, bad, WGNN_STATUS_ACT_RANGE);
}

// One backward layer:
//   dz = dOut * (Out > 0)           where dOut is given directly (layer 2: dg) or as A^T DU (layer 1: gathered like A In)
//   p  = (A In)[s]                  (recomputed aggregation)
//   dW += p^T dz, db += dz          on v_mfma_f32_16x16x4_f32: the rows' p (with a ones column: db) and dz go through LDS as
//                                   [row][16] and every wave contracts its 64 rows in 16 MFMAs; the fp32 accumulator tile is
//                                   added into fp64 registers after every 64 rows (these are long sums of mixed-sign terms,
//                                   cancellation ~100x at S = 4096) and reduced over the block's waves at the end
//   DU[tile][s][:] = dz W^T         (layer 2 only: what layer 1's A^T product consumes)
template <bool LAYER2>
__global__ void __launch_bounds__(CT) csr_layer_bwd_kernel(int ntiles, int S, Csr A, Csr AT,
                                                           const float* __restrict__ In,
                                                           const float* __restrict__ Out,
                                                           const _Float16* __restrict__ Out_hi, size_t ld_out,
                                                           const float* __restrict__ dOut, size_t ld_dout,
                                                           const float* __restrict__ W, float* __restrict__ DU,
                                                           const float* __restrict__ scales,
                                                           float* __restrict__ partial) {
  constexpr int RPT = 2, RB = CT * RPT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile = lds;
  float* Ws = lds + CH * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (LAYER2) {
    for (int i = tid; i < F13 * F13; i += CT) Ws[i] = W[i];
    __syncthreads();
  }
  const int I = S * F13;
  double acc64[4] = {0.0, 0.0, 0.0, 0.0};      // lane (c = lane & 15, g = lane >> 4), r: dW[f = 4 g + r][c]; f = 13: db[c]
  const int nitems = csr_items(ntiles, S, RB);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const CsrItem it = csr_item(item, ntiles, S, RB);
    int e0[RPT], e1[RPT], coff[RPT];
    float dz[RPT][F13], p[RPT][F13];
    if (!LAYER2) {                                             // dH1 = A^T DU (DU = dOut, [ntiles][S*13])
      csr_rows<RPT>(AT, it, S, e0, e1, coff);
      csr_gather<RPT, 2>(AT, dOut + (size_t)it.tile0 * I, it.VC, e0, e1, coff, tile, dz);
    }
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      int g, srow;
      if (!csr_row_of(it, S, q, g, srow)) {
#pragma unroll
        for (int f = 0; f < F13; ++f) dz[q][f] = 0.f;
        continue;
      }
      const size_t oo = (size_t)(it.tile0 + g) * ld_out + (size_t)srow * F13;
      bool on[F13];                                            // ReLU mask: fp32 activations or their fp16 hi plane
#pragma unroll
      for (int f = 0; f < F13; ++f) on[f] = Out_hi ? (float)Out_hi[oo + f] > 0.f : Out[oo + f] > 0.f;
      if (LAYER2) {
        const float* d = dOut + (size_t)(it.tile0 + g) * ld_dout + (size_t)srow * F13;
#pragma unroll
        for (int f = 0; f < F13; ++f) dz[q][f] = on[f] ? d[f] : 0.f;
      } else {
#pragma unroll
        for (int f = 0; f < F13; ++f) dz[q][f] = on[f] ? dz[q][f] : 0.f;
      }
    }
    if (LAYER2) {
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        int g, srow;
        if (!csr_row_of(it, S, q, g, srow)) continue;
        float* du = DU + ((size_t)(it.tile0 + g) * S + srow) * F13;
#pragma unroll
        for (int f = 0; f < F13; ++f) {
          float a = 0.f;
#pragma unroll
          for (int c = 0; c < F13; ++c) a = fmaf(dz[q][c], Ws[f * F13 + c], a);
          du[f] = a;
        }
      }
    }
    csr_rows<RPT>(A, it, S, e0, e1, coff);
    csr_gather<RPT, 2>(A, In + (size_t)it.tile0 * I, it.VC, e0, e1, coff, tile, p);
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      if (q * CT >= it.nrows) break;                           // block-uniform
      __syncthreads();                                         // the chunk / the previous rows' operands have been read
      f32x4* P = (f32x4*)(tile + (size_t)tid * 16);
      f32x4* Z = (f32x4*)(tile + (size_t)(CT + tid) * 16);
      const bool live = tid + CT * q < it.nrows;                 // rows past the item: p = dz = 0 too
      const float one = live ? 1.f : 0.f;
      P[0] = f32x4{live ? p[q][0] : 0.f, live ? p[q][1] : 0.f, live ? p[q][2] : 0.f, live ? p[q][3] : 0.f};
      P[1] = f32x4{live ? p[q][4] : 0.f, live ? p[q][5] : 0.f, live ? p[q][6] : 0.f, live ? p[q][7] : 0.f};
      P[2] = f32x4{live ? p[q][8] : 0.f, live ? p[q][9] : 0.f, live ? p[q][10] : 0.f, live ? p[q][11] : 0.f};
      P[3] = f32x4{live ? p[q][12] : 0.f, one, 0.f, 0.f};
      Z[0] = f32x4{dz[q][0], dz[q][1], dz[q][2], dz[q][3]};
      Z[1] = f32x4{dz[q][4], dz[q][5], dz[q][6], dz[q][7]};
      Z[2] = f32x4{dz[q][8], dz[q][9], dz[q][10], dz[q][11]};
      Z[3] = f32x4{dz[q][12], 0.f, 0.f, 0.f};
      __syncthreads();
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* pr = tile + (size_t)(64 * wave + (lane >> 4)) * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc = mfma16(pr[j * 64], pr[CT * 16 + j * 64], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc64[r] += (double)acc[r];
    }
  }
  // ---- block reduction in wave order, one partial row per block
  __syncthreads();
  double* red = (double*)lds;
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(size_t)wave * 256 + lane * 4 + r] = acc64[r];
  __syncthreads();
  if (tid < 256) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < CT / 64; ++w) t += red[w * 256 + tid];
    if (scales) t *= (double)scales[1];      // dOut arrives in the backward's power-of-two scaled units
    const int l = tid >> 2, r = tid & 3, f = 4 * (l >> 4) + r, c = l & 15;
    float* mine = partial + (size_t)blockIdx.x * PART;
    if (c < F13) {
      if (f < F13) mine[(LAYER2 ? FP * FP : 0) + f * FP + c] = (float)t;
      else if (f == F13) mine[2 * FP * FP + (LAYER2 ? FP : 0) + c] = (float)t;
    }
  }
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

size_t gcn_csr_bwd_partial_floats() { return (size_t)CSR_BLOCKS * PART; }

// g: fp32 [ntiles][ldg], or (g_planes != nullptr) fp16 hi/lo planes [ntiles][ldg] each (lo skipped if !x3)
// opt the three LDS-tiled kernels into their 129 KB of dynamic LDS (once per device)
static int csr_lds_ready() {
  static std::atomic<unsigned long long> d0{0}, d1{0}, d2{0}, d3{0};
  int rc = ensure_dyn_smem((const void*)csr_layer_fwd_kernel<false>, CSR_LDS, d0);
  if (rc == WGNN_OK) rc = ensure_dyn_smem((const void*)csr_layer_fwd_kernel<true>, CSR_LDS, d1);
  if (rc == WGNN_OK) rc = ensure_dyn_smem((const void*)csr_layer_bwd_kernel<false>, CSR_LDS, d2);
  if (rc == WGNN_OK) rc = ensure_dyn_smem((const void*)csr_layer_bwd_kernel<true>, CSR_LDS, d3);
  return rc;
}
static int csr_grid(int ntiles, int S, int rpt) {
  const int n = csr_items(ntiles, S, CT * rpt);
  return n < CSR_BLOCKS ? (n < 1 ? 1 : n) : CSR_BLOCKS;
}

int launch_gcn2_csr_fwd(int ntiles, int S, int nnz, const void* csr, const float* X, const float* W1, const float* b1,
                        const float* W2, const float* b2, float* h1, float* g, void* g_planes, size_t ldg, bool x3,
                        unsigned* status, hipStream_t st) {
  const Csr A = csr_of(csr, S, nnz, false);
  if (csr_lds_ready() != WGNN_OK) return WGNN_ERR_HIP;
  const dim3 grid(csr_grid(ntiles, S, CSR_FWD_RPT));
  const double fl = (double)ntiles * (2.0 * nnz * F13 + 2.0 * S * F13 * F13), by = (double)ntiles * S * F13 * 8.0;
  const size_t I = (size_t)S * F13;
  PROF_LAUNCH("csr_layer_fwd_kernel", fl, by, st,
              hipLaunchKernelGGL(csr_layer_fwd_kernel<false>, grid, dim3(CT), CSR_LDS, st, ntiles, S, A, X, W1, b1, h1, I,
                                 (_Float16*)nullptr, (_Float16*)nullptr, (unsigned*)nullptr));
  WGNN_CHECK_LAUNCH();
  if (g_planes) {
    _Float16* hi = (_Float16*)g_planes;
    PROF_LAUNCH("csr_layer_fwd_kernel", fl, by, st,
                hipLaunchKernelGGL(csr_layer_fwd_kernel<true>, grid, dim3(CT), CSR_LDS, st, ntiles, S, A, h1, W2, b2,
                                   (float*)nullptr, ldg, hi, x3 ? hi + (size_t)ntiles * ldg : (_Float16*)nullptr, status));
  } else {
    PROF_LAUNCH("csr_layer_fwd_kernel", fl, by, st,
                hipLaunchKernelGGL(csr_layer_fwd_kernel<false>, grid, dim3(CT), CSR_LDS, st, ntiles, S, A, h1, W2, b2, g,
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
  if (csr_lds_ready() != WGNN_OK) return WGNN_ERR_HIP;
  // every block writes its partial row (zeros if it had no item): always CSR_BLOCKS blocks
  PROF_LAUNCH("csr_layer_bwd_kernel<2>", fl, by, st,
              hipLaunchKernelGGL((csr_layer_bwd_kernel<true>), dim3(CSR_BLOCKS), dim3(CT), CSR_LDS, st, ntiles, S, A, AT, h1,
                                 g, (const _Float16*)g_hi, ldg, dg, ld_dg, W2, du, scales, partial));
  WGNN_CHECK_LAUNCH();
  PROF_LAUNCH("csr_layer_bwd_kernel<1>", fl + (double)ntiles * 2.0 * nnz * F13, by, st,
              hipLaunchKernelGGL((csr_layer_bwd_kernel<false>), dim3(CSR_BLOCKS), dim3(CT), CSR_LDS, st, ntiles, S, A, AT, X,
                                 h1, (const _Float16*)nullptr, I, du, I, (const float*)nullptr, (float*)nullptr,
                                 scales, partial));
  WGNN_CHECK_LAUNCH();
  if (!dW1) return WGNN_OK;              // deferred: finish.hip reduces the gcn_csr_bwd_rows() partial rows
  return launch_gcn_partial_reduce(partial, CSR_BLOCKS, dW1, db1, dW2, db2, nullptr, st);
}

int gcn_csr_bwd_rows() { return CSR_BLOCKS; }

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
  if (csr_lds_ready() != WGNN_OK) return WGNN_ERR_HIP;
  const size_t I = (size_t)S * F13;
  hipLaunchKernelGGL(csr_layer_fwd_kernel<false>, dim3(csr_grid(ntiles, S, CSR_FWD_RPT)), dim3(CT), CSR_LDS, st, ntiles, S, A, X, W, b,
                     out, I, (_Float16*)nullptr, (_Float16*)nullptr, (unsigned*)nullptr);
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
  if (csr_lds_ready() != WGNN_OK) return WGNN_ERR_HIP;
  hipLaunchKernelGGL((csr_layer_bwd_kernel<true>), dim3(CSR_BLOCKS), dim3(CT), CSR_LDS, st, ntiles, S, A, AT, X, out,
                     (const _Float16*)nullptr, I, dout, I, W, du, (const float*)nullptr, partial);
  WGNN_CHECK_LAUNCH();
  int rc = launch_gcn_partial_reduce(partial, CSR_BLOCKS, nullptr, nullptr, dW, db, nullptr, st);   // the "layer 2" slots
  if (rc != WGNN_OK || !dX) return rc;
  const dim3 grid(cdiv_i(S, ROWS), ntiles < 16384 ? ntiles : 16384);
  hipLaunchKernelGGL(csr_spmm_kernel, grid, dim3(ROWS), 0, st, ntiles, S, AT, du, dX);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
