// Graph-convolution kernels (dense adjacency, S <= 64), exact-fp32 MFMA.
//
// Reference: GraphConvLayer.forward, src/step5_gcn_layer_model.py:13-23  out = relu((A X) W + b)
// and its two stacked uses in GCN_GRU.forward, src/step6_gcn_gru_combined_model.py:17-20.
//
// One wavefront owns one (window, timestep) tile X_t [S,13] end to end; a workgroup of 4 waves
// shares the normalised adjacency (and its transpose, augmented with a row of ones that yields the
// bias gradient for free) in LDS.  Every product is a chain of v_mfma_f32_16x16x4_f32 whose
// operands are read from small per-wave LDS tiles, so any orientation (A, A^T, W, W^T, X^T) is
// just index arithmetic; results are bitwise an fp32 fmaf chain.
//
// HBM traffic per tile: forward reads X (S*13*4 B) and writes g (same); backward reads X, g, dg.
#include "common.h"

namespace {

constexpr int FP = 16;   // feature dim padded to one MFMA tile
constexpr int RS = 18;   // row stride (floats) of the per-wave [rows][16] LDS tiles: conflict-light
constexpr int WAVES = 4;

struct GcnGeom {
  int S, NT, NTa, KS, AST, SP, SPa;
  __host__ __device__ explicit GcnGeom(int S_) {
    S = S_;
    NT = (S + 15) / 16;
    NTa = (S + 1 + 15) / 16;   // room for the augmented ones-row at index S
    KS = (S + 3) / 4;
    AST = 4 * KS + 2;          // == 2 (mod 4): A-operand reads hit 32 distinct banks
    SP = 16 * NT;
    SPa = 16 * NTa;
  }
  __host__ __device__ int shared_floats() const { return SP * AST + SPa * AST + 2 * FP * FP + 2 * FP; }
  __host__ __device__ int wave_floats() const { return 4 * SPa * RS; }
};

struct Lane {
  int lane, lm, lk;
  __device__ Lane() {
    lane = threadIdx.x & 63;
    lm = lane & 15;
    lk = lane >> 4;
  }
};

// acc(16x16) per m-tile = sum_k fa(mt,ks) * fb(ks); fe(mt, acc) consumes it.
template <class FA, class FB, class FE>
__device__ __forceinline__ void mm_stage(int nmt, int nks, FA fa, FB fb, FE fe) {
  for (int mt = 0; mt < nmt; ++mt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < nks; ++ks) acc = mfma16(fa(mt, ks), fb(ks), acc);
    fe(mt, acc);
  }
}

__device__ __forceinline__ void store_tile(float* T, int mt, const Lane& L, f32x4 acc) {
#pragma unroll
  for (int r = 0; r < 4; ++r) T[(16 * mt + 4 * L.lk + r) * RS + L.lm] = acc[r];
}

// shared-LDS setup: A (padded), A^T augmented with ones-row S, W1, W2 (16x16 zero padded), b1, b2
__device__ void load_shared(const GcnGeom& G, float* As, float* ATs, float* W1s, float* W2s,
                            float* b1s, float* b2s, const float* A, const float* W1,
                            const float* b1, const float* W2, const float* b2, int F) {
  const int S = G.S;
  for (int i = threadIdx.x; i < G.SP * G.AST; i += blockDim.x) {
    int r = i / G.AST, c = i % G.AST;
    As[i] = (r < S && c < S) ? A[r * S + c] : 0.f;
  }
  for (int i = threadIdx.x; i < G.SPa * G.AST; i += blockDim.x) {
    int r = i / G.AST, c = i % G.AST;
    float v = 0.f;
    if (c < S) v = (r < S) ? A[c * S + r] : (r == S ? 1.f : 0.f);
    ATs[i] = v;
  }
  for (int i = threadIdx.x; i < FP * FP; i += blockDim.x) {
    int r = i / FP, c = i % FP;
    bool in = r < F && c < F;
    W1s[i] = (in && W1) ? W1[r * F + c] : 0.f;
    W2s[i] = (in && W2) ? W2[r * F + c] : 0.f;
  }
  for (int i = threadIdx.x; i < FP; i += blockDim.x) {
    b1s[i] = (i < F && b1) ? b1[i] : 0.f;
    b2s[i] = (i < F && b2) ? b2[i] : 0.f;
  }
}

__device__ __forceinline__ void zero_wave_tiles(float* Wv, int n, int lane) {
  for (int i = lane; i < n; i += 64) Wv[i] = 0.f;
}

// linear [S*F] global tile -> Ts[s][f] (pads untouched, they stay zero)
__device__ __forceinline__ void load_tile(float* Ts, const float* src, int S, int F, int lane, bool valid) {
  const int n = S * F;
  for (int i = lane; i < n; i += 64) {
    float v = valid ? src[i] : 0.f;
    Ts[(i / F) * RS + (i % F)] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// two-layer forward: g = relu(A relu(A X W1 + b1) W2 + b2), flattened [S*F] per tile
// LAYERS == 1: out = relu(A X W1 + b1)
template <int LAYERS>
__global__ void __launch_bounds__(256) gcn_fwd_kernel(int ntiles, int S, int F, const float* __restrict__ A,
                                                      const float* __restrict__ X, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, const float* __restrict__ W2,
                                                      const float* __restrict__ b2, float* __restrict__ out,
                                                      int ld_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GcnGeom G(S);
  float* As = smem;
  float* ATs = As + G.SP * G.AST;
  float* W1s = ATs + G.SPa * G.AST;
  float* W2s = W1s + FP * FP;
  float* b1s = W2s + FP * FP;
  float* b2s = b1s + FP;
  const int wave = threadIdx.x >> 6;
  float* Wv = b2s + FP + wave * G.wave_floats();
  float* Xs = Wv;
  float* Us = Xs + G.SPa * RS;
  float* Hs = Us + G.SPa * RS;
  const Lane L;
  load_shared(G, As, ATs, W1s, W2s, b1s, b2s, A, W1, b1, W2, b2, F);
  zero_wave_tiles(Wv, G.wave_floats(), L.lane);
  __syncthreads();
  const int I = S * F;
  const float bias1 = b1s[L.lm], bias2 = b2s[L.lm];

  for (int base = blockIdx.x * WAVES; base < ntiles; base += gridDim.x * WAVES) {
    const int tile = base + wave;
    const bool valid = tile < ntiles;
    load_tile(Xs, X + (size_t)(valid ? tile : 0) * I, S, F, L.lane, valid);
    __syncthreads();
    // U = X W1
    mm_stage(G.NT, 4, [&](int mt, int ks) { return Xs[(16 * mt + L.lm) * RS + 4 * ks + L.lk]; },
             [&](int ks) { return W1s[(4 * ks + L.lk) * FP + L.lm]; },
             [&](int mt, f32x4 acc) { store_tile(Us, mt, L, acc); });
    __syncthreads();
    // H1 = relu(A U + b1), rows >= S forced to 0
    mm_stage(G.NT, G.KS, [&](int mt, int ks) { return As[(16 * mt + L.lm) * G.AST + 4 * ks + L.lk]; },
             [&](int ks) { return Us[(4 * ks + L.lk) * RS + L.lm]; },
             [&](int mt, f32x4 acc) {
#pragma unroll
               for (int r = 0; r < 4; ++r) {
                 int row = 16 * mt + 4 * L.lk + r;
                 float v = fmaxf(acc[r] + bias1, 0.f);
                 Hs[row * RS + L.lm] = row < S ? v : 0.f;
               }
             });
    __syncthreads();
    if (LAYERS == 2) {
      mm_stage(G.NT, 4, [&](int mt, int ks) { return Hs[(16 * mt + L.lm) * RS + 4 * ks + L.lk]; },
               [&](int ks) { return W2s[(4 * ks + L.lk) * FP + L.lm]; },
               [&](int mt, f32x4 acc) { store_tile(Us, mt, L, acc); });
      __syncthreads();
      mm_stage(G.NT, G.KS, [&](int mt, int ks) { return As[(16 * mt + L.lm) * G.AST + 4 * ks + L.lk]; },
               [&](int ks) { return Us[(4 * ks + L.lk) * RS + L.lm]; },
               [&](int mt, f32x4 acc) {
#pragma unroll
                 for (int r = 0; r < 4; ++r) {
                   int row = 16 * mt + 4 * L.lk + r;
                   float v = fmaxf(acc[r] + bias2, 0.f);
                   Hs[row * RS + L.lm] = row < S ? v : 0.f;
                 }
               });
      __syncthreads();
    }
    if (valid) {
      float* dst = out + (size_t)tile * ld_out;
      for (int i = L.lane; i < I; i += 64) dst[i] = Hs[(i / F) * RS + (i % F)];
      for (int i = I + L.lane; i < ld_out; i += 64) dst[i] = 0.f;   // K padding of the projection GEMM
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// backward.  LAYERS == 2: inputs X, g (= layer-2 output, the relu mask), dg; outputs per-workgroup
// partial sums of dW1, db1, dW2, db2 (no dX: the path's input does not require grad).
// LAYERS == 1: inputs X, out, dout; partials of dW, db (slots of layer 1) and optional dX.
// partial layout per workgroup: [dW1 16x16 | dW2 16x16 | db1 16 | db2 16] = 544 floats.
constexpr int PART = 2 * FP * FP + 2 * FP;

template <int LAYERS>
__global__ void __launch_bounds__(256) gcn_bwd_kernel(int ntiles, int S, int F, const float* __restrict__ A,
                                                      const float* __restrict__ X, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, const float* __restrict__ W2,
                                                      const float* __restrict__ gout, const float* __restrict__ dgout,
                                                      float* __restrict__ dX, float* __restrict__ partial,
                                                      int ld_g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GcnGeom G(S);
  float* As = smem;
  float* ATs = As + G.SP * G.AST;
  float* W1s = ATs + G.SPa * G.AST;
  float* W2s = W1s + FP * FP;
  float* b1s = W2s + FP * FP;
  float* b2s = b1s + FP;
  const int wave = threadIdx.x >> 6;
  float* Wv = b2s + FP + wave * G.wave_floats();
  float* Xs = Wv;
  float* Us = Xs + G.SPa * RS;
  float* Hs = Us + G.SPa * RS;
  float* Ds = Hs + G.SPa * RS;
  const Lane L;
  load_shared(G, As, ATs, W1s, W2s, b1s, b2s, A, W1, b1, W2, nullptr, F);
  zero_wave_tiles(Wv, G.wave_floats(), L.lane);
  __syncthreads();
  const int I = S * F;
  const float bias1 = b1s[L.lm];
  // location of the augmented ones-row S inside the C layout
  const int aug_mt = S / 16, aug_lk = (S % 16) / 4, aug_r = S % 4;

  f32x4 dW1acc = {0.f, 0.f, 0.f, 0.f}, dW2acc = {0.f, 0.f, 0.f, 0.f};
  float db1acc = 0.f, db2acc = 0.f;

  for (int base = blockIdx.x * WAVES; base < ntiles; base += gridDim.x * WAVES) {
    const int tile = base + wave;
    const bool valid = tile < ntiles;
    const size_t off = (size_t)(valid ? tile : 0) * I;
    load_tile(Xs, X + off, S, F, L.lane, valid);
    const size_t goff = (size_t)(valid ? tile : 0) * ld_g;
    for (int i = L.lane; i < I; i += 64) {   // dZ_last = dout * (out > 0)
      float v = (valid && gout[goff + i] > 0.f) ? dgout[off + i] : 0.f;
      Ds[(i / F) * RS + (i % F)] = v;
    }
    __syncthreads();
    if (LAYERS == 2) {
      // recompute H1 = relu(A (X W1) + b1)
      mm_stage(G.NT, 4, [&](int mt, int ks) { return Xs[(16 * mt + L.lm) * RS + 4 * ks + L.lk]; },
               [&](int ks) { return W1s[(4 * ks + L.lk) * FP + L.lm]; },
               [&](int mt, f32x4 acc) { store_tile(Us, mt, L, acc); });
      __syncthreads();
      mm_stage(G.NT, G.KS, [&](int mt, int ks) { return As[(16 * mt + L.lm) * G.AST + 4 * ks + L.lk]; },
               [&](int ks) { return Us[(4 * ks + L.lk) * RS + L.lm]; },
               [&](int mt, f32x4 acc) {
#pragma unroll
                 for (int r = 0; r < 4; ++r) {
                   int row = 16 * mt + 4 * L.lk + r;
                   float v = fmaxf(acc[r] + bias1, 0.f);
                   Hs[row * RS + L.lm] = row < S ? v : 0.f;
                 }
               });
      __syncthreads();
      // dU2 = [A^T ; 1^T] dZ2  (row S = column sums of dZ2 = db2 contribution)
      mm_stage(G.NTa, G.KS, [&](int mt, int ks) { return ATs[(16 * mt + L.lm) * G.AST + 4 * ks + L.lk]; },
               [&](int ks) { return Ds[(4 * ks + L.lk) * RS + L.lm]; },
               [&](int mt, f32x4 acc) {
                 store_tile(Us, mt, L, acc);
                 if (mt == aug_mt && L.lk == aug_lk) db2acc += acc[aug_r];
               });
      __syncthreads();
      // dW2 += H1^T dU2   (contraction over stations; H1 rows >= S are zero)
      for (int ks = 0; ks < G.KS; ++ks)
        dW2acc = mfma16(Hs[(4 * ks + L.lk) * RS + L.lm], Us[(4 * ks + L.lk) * RS + L.lm], dW2acc);
      // dZ1 = (dU2 W2^T) * (H1 > 0)
      mm_stage(G.NT, 4, [&](int mt, int ks) { return Us[(16 * mt + L.lm) * RS + 4 * ks + L.lk]; },
               [&](int ks) { return W2s[L.lm * FP + 4 * ks + L.lk]; },
               [&](int mt, f32x4 acc) {
#pragma unroll
                 for (int r = 0; r < 4; ++r) {
                   int idx = (16 * mt + 4 * L.lk + r) * RS + L.lm;
                   Ds[idx] = Hs[idx] > 0.f ? acc[r] : 0.f;
                 }
               });
      __syncthreads();
    }
    // dU1 = [A^T ; 1^T] dZ1
    mm_stage(G.NTa, G.KS, [&](int mt, int ks) { return ATs[(16 * mt + L.lm) * G.AST + 4 * ks + L.lk]; },
             [&](int ks) { return Ds[(4 * ks + L.lk) * RS + L.lm]; },
             [&](int mt, f32x4 acc) {
               store_tile(Us, mt, L, acc);
               if (mt == aug_mt && L.lk == aug_lk) db1acc += acc[aug_r];
             });
    __syncthreads();
    // dW1 += X^T dU1
    for (int ks = 0; ks < G.KS; ++ks)
      dW1acc = mfma16(Xs[(4 * ks + L.lk) * RS + L.lm], Us[(4 * ks + L.lk) * RS + L.lm], dW1acc);
    if (LAYERS == 1 && dX != nullptr) {
      // dX = dU1 W1^T
      mm_stage(G.NT, 4, [&](int mt, int ks) { return Us[(16 * mt + L.lm) * RS + 4 * ks + L.lk]; },
               [&](int ks) { return W1s[L.lm * FP + 4 * ks + L.lk]; },
               [&](int mt, f32x4 acc) { store_tile(Hs, mt, L, acc); });
      __syncthreads();
      if (valid)
        for (int i = L.lane; i < I; i += 64) dX[off + i] = Hs[(i / F) * RS + (i % F)];
    }
    __syncthreads();
  }

  // reduce the 4 waves' accumulators through LDS (reuse wave 0's tiles) and emit one partial row
  __syncthreads();
  float* red = b2s + FP;   // start of per-wave area; WAVES * PART floats fit easily
  float* mine = red + wave * PART;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    mine[(4 * L.lk + r) * FP + L.lm] = dW1acc[r];
    mine[FP * FP + (4 * L.lk + r) * FP + L.lm] = dW2acc[r];
  }
  if (L.lk == aug_lk) {
    mine[2 * FP * FP + L.lm] = db1acc;
    mine[2 * FP * FP + FP + L.lm] = db2acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PART; i += blockDim.x) {
    float s = 0.f;
    for (int w = 0; w < WAVES; ++w) s += red[w * PART + i];
    partial[(size_t)blockIdx.x * PART + i] = s;
  }
}

// sum partial rows (fixed order => bitwise reproducible) and scatter into the [F,F]/[F] gradients.
// One block per 32 columns; 32 row-groups per block (1024 threads) keep many independent loads in flight.
__global__ void __launch_bounds__(1024) gcn_partial_reduce_kernel(const float* __restrict__ partial, int nblk, int F,
                                                                  float* dW1, float* db1, float* dW2, float* db2,
                                                                  unsigned* status) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < PART) {
    int b = ty;
    for (; b + 96 < nblk; b += 128) {
      s0 += partial[(size_t)b * PART + i];
      s1 += partial[(size_t)(b + 32) * PART + i];
      s2 += partial[(size_t)(b + 64) * PART + i];
      s3 += partial[(size_t)(b + 96) * PART + i];
    }
    for (; b < nblk; b += 32) s0 += partial[(size_t)b * PART + i];
  }
  sm[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty != 0 || i >= PART) return;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) s += sm[k][tx];
  report_status(status, !(__builtin_fabsf(s) <= 3.0e38f), WGNN_STATUS_GRAD_NONFINITE);
  if (i < FP * FP) {
    int r = i / FP, c = i % FP;
    if (dW1 && r < F && c < F) dW1[r * F + c] = s;
  } else if (i < 2 * FP * FP) {
    int j = i - FP * FP, r = j / FP, c = j % FP;
    if (dW2 && r < F && c < F) dW2[r * F + c] = s;
  } else if (i < 2 * FP * FP + FP) {
    int c = i - 2 * FP * FP;
    if (db1 && c < F) db1[c] = s;
  } else {
    int c = i - 2 * FP * FP - FP;
    if (db2 && c < F) db2[c] = s;
  }
}

int grid_for(int ntiles) {
  int g = cdiv_i(ntiles, WAVES);
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

size_t smem_bytes(int S) {
  GcnGeom G(S);
  return (size_t)(G.shared_floats() + WAVES * G.wave_floats()) * sizeof(float);
}

}  // namespace

int launch_gcn_partial_reduce(const float* partial, int nblk, float* dW1, float* db1, float* dW2, float* db2,
                              unsigned* status, hipStream_t st) {
  PROF_LAUNCH("gcn_partial_reduce_kernel", (double)nblk * PART, 4.0 * nblk * PART, st,
              hipLaunchKernelGGL(gcn_partial_reduce_kernel, dim3(cdiv_i(PART, 32)), dim3(1024), 0, st, partial, nblk,
                                 13, dW1, db1, dW2, db2, status));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

size_t gcn1_bwd_partial_floats(int ntiles) { return (size_t)grid_for(ntiles) * PART; }

int launch_gcn1_fwd(int ntiles, int S, const float* A, const float* X, const float* W, const float* b, float* out,
                    hipStream_t st) {
  hipLaunchKernelGGL(gcn_fwd_kernel<1>, dim3(grid_for(ntiles)), dim3(256), smem_bytes(S), st, ntiles, S, 13, A, X, W,
                     b, (const float*)nullptr, (const float*)nullptr, out, S * 13);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_gcn1_bwd(int ntiles, int S, const float* A, const float* X, const float* W, const float* out,
                    const float* dout, float* dW, float* db, float* dX, float* partial, hipStream_t st) {
  int grid = grid_for(ntiles);
  hipLaunchKernelGGL(gcn_bwd_kernel<1>, dim3(grid), dim3(256), smem_bytes(S), st, ntiles, S, 13, A, X, W,
                     (const float*)nullptr, (const float*)nullptr, out, dout, dX, partial, S * 13);
  WGNN_CHECK_LAUNCH();
  hipLaunchKernelGGL(gcn_partial_reduce_kernel, dim3(cdiv_i(PART, 32)), dim3(1024), 0, st, partial, grid, 13, dW,
                     db, (float*)nullptr, (float*)nullptr, (unsigned*)nullptr);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
