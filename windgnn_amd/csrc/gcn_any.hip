// GraphConvLayer with ANY feature widths (VERDICT r4 missing 3 / next 8).
//
// Reference: GraphConvLayer(input_dim, output_dim), src/step5_gcn_layer_model.py:6-23 -- weight [input_dim, output_dim], bias
// [output_dim], forward relu((A X) W + b) for X [..., S, input_dim].  The reference's own model only ever builds 13 -> 13
// layers (src/main.py:41), and those run the MFMA kernels of gcn.hip / gcn32.hip / gcnx.hip; this file is the rest of the
// constructor's domain: F_in, F_out <= 64 over a dense adjacency of S <= 64 stations, in exact fp32 (fmaf chains in k order).
// It is not a hot path (nothing in BASELINE.json runs it): one 256-thread workgroup per tile at a time, A / W / b resident in
// LDS, the tile's X and A X staged in LDS, plain FMAs -- HBM-bound at these widths (reads X, writes out; backward reads X, out,
// dout and writes dX).  Weight gradients: per-workgroup partial sums in registers over the workgroup's tiles (thread t owns the
// (k, f) pairs t, t + 256, ...), written once per workgroup and summed by a second launch in fixed order (bitwise reproducible).
#include "common.h"

namespace {

constexpr int GA_THREADS = 256;
constexpr int GA_MAXF = 64, GA_MAXS = 64;
constexpr int GA_PAIRS = GA_MAXF * GA_MAXF / GA_THREADS;          // (k, f) pairs per thread at the widest layer: 16

struct GaGeom {
  int S, Fi, Fo;
  __host__ __device__ size_t smem_floats(bool bwd) const {
    // A | W | b | X tile | P = A X tile | (bwd) dZ tile | (bwd) dU tile
    return (size_t)S * S + (size_t)Fi * Fo + Fo + 2 * (size_t)S * Fi + (bwd ? (size_t)S * Fo + (size_t)S * Fi : 0);
  }
};

__device__ __forceinline__ void ga_load_consts(const GaGeom g, const float* A, const float* W, const float* b, float* sA, float* sW,
                                               float* sb) {
  for (int i = threadIdx.x; i < g.S * g.S; i += GA_THREADS) sA[i] = A[i];
  for (int i = threadIdx.x; i < g.Fi * g.Fo; i += GA_THREADS) sW[i] = W[i];
  for (int i = threadIdx.x; i < g.Fo; i += GA_THREADS) sb[i] = b ? b[i] : 0.f;
}

// P[s][k] = sum_j A[s][j] X[j][k]   (src/step5_gcn_layer_model.py:15)
__device__ __forceinline__ void ga_aggregate(const GaGeom g, const float* sA, const float* sX, float* sP) {
  for (int i = threadIdx.x; i < g.S * g.Fi; i += GA_THREADS) {
    const int s = i / g.Fi, k = i % g.Fi;
    float acc = 0.f;
    for (int j = 0; j < g.S; ++j) acc = __builtin_fmaf(sA[s * g.S + j], sX[j * g.Fi + k], acc);
    sP[i] = acc;
  }
}

__global__ void __launch_bounds__(GA_THREADS) gcn_any_fwd_kernel(GaGeom g, int ntiles, const float* __restrict__ A,
                                                                const float* __restrict__ X, const float* __restrict__ W,
                                                                const float* __restrict__ b, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sA = sm;
  float* sW = sA + g.S * g.S;
  float* sb = sW + g.Fi * g.Fo;
  float* sX = sb + g.Fo;
  float* sP = sX + g.S * g.Fi;
  ga_load_consts(g, A, W, b, sA, sW, sb);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();                                              // constants loaded / the previous tile's P no longer read
    const float* x = X + (size_t)t * g.S * g.Fi;
    for (int i = threadIdx.x; i < g.S * g.Fi; i += GA_THREADS) sX[i] = x[i];
    __syncthreads();
    ga_aggregate(g, sA, sX, sP);
    __syncthreads();
    float* o = out + (size_t)t * g.S * g.Fo;
    for (int i = threadIdx.x; i < g.S * g.Fo; i += GA_THREADS) {  // relu(P W + b)   (:18, :21)
      const int s = i / g.Fo, f = i % g.Fo;
      float acc = 0.f;
      for (int k = 0; k < g.Fi; ++k) acc = __builtin_fmaf(sP[s * g.Fi + k], sW[k * g.Fo + f], acc);
      acc += sb[f];
      o[i] = acc > 0.f ? acc : 0.f;
    }
  }
}

// Backward of one layer: dZ = dout * [out > 0]; dW += P^T dZ; db += sum_s dZ; dX = A^T (dZ W^T) (if wanted).
// partial: [gridDim.x][Fi * Fo + Fo] per-workgroup sums.
__global__ void __launch_bounds__(GA_THREADS) gcn_any_bwd_kernel(GaGeom g, int ntiles, const float* __restrict__ A,
                                                                const float* __restrict__ X, const float* __restrict__ W,
                                                                const float* __restrict__ out, const float* __restrict__ dout,
                                                                float* __restrict__ dX, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sA = sm;
  float* sW = sA + g.S * g.S;
  float* sb = sW + g.Fi * g.Fo;
  float* sX = sb + g.Fo;
  float* sP = sX + g.S * g.Fi;
  float* sZ = sP + g.S * g.Fi;
  float* sU = sZ + g.S * g.Fo;
  ga_load_consts(g, A, W, nullptr, sA, sW, sb);
  float accW[GA_PAIRS];
#pragma unroll
  for (int q = 0; q < GA_PAIRS; ++q) accW[q] = 0.f;
  float accb = 0.f;                                               // thread f < Fo owns db[f]
  const int npair = g.Fi * g.Fo;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();
    const float* x = X + (size_t)t * g.S * g.Fi;
    for (int i = threadIdx.x; i < g.S * g.Fi; i += GA_THREADS) sX[i] = x[i];
    const float* o = out + (size_t)t * g.S * g.Fo;
    const float* dz = dout + (size_t)t * g.S * g.Fo;
    for (int i = threadIdx.x; i < g.S * g.Fo; i += GA_THREADS) sZ[i] = o[i] > 0.f ? dz[i] : 0.f;
    __syncthreads();
    ga_aggregate(g, sA, sX, sP);
    if (dX) {                                                     // dU[s][k] = sum_f dZ[s][f] W[k][f]
      for (int i = threadIdx.x; i < g.S * g.Fi; i += GA_THREADS) {
        const int s = i / g.Fi, k = i % g.Fi;
        float acc = 0.f;
        for (int f = 0; f < g.Fo; ++f) acc = __builtin_fmaf(sZ[s * g.Fo + f], sW[k * g.Fo + f], acc);
        sU[i] = acc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < GA_PAIRS; ++q) {                          // dW[k][f] += sum_s P[s][k] dZ[s][f]
      const int pr = threadIdx.x + q * GA_THREADS;
      if (pr < npair) {
        const int k = pr / g.Fo, f = pr % g.Fo;
        float acc = accW[q];
        for (int s = 0; s < g.S; ++s) acc = __builtin_fmaf(sP[s * g.Fi + k], sZ[s * g.Fo + f], acc);
        accW[q] = acc;
      }
    }
    if ((int)threadIdx.x < g.Fo)
      for (int s = 0; s < g.S; ++s) accb += sZ[s * g.Fo + threadIdx.x];
    if (dX) {                                                     // dX[j][k] = sum_s A[s][j] dU[s][k]
      float* dx = dX + (size_t)t * g.S * g.Fi;
      for (int i = threadIdx.x; i < g.S * g.Fi; i += GA_THREADS) {
        const int j = i / g.Fi, k = i % g.Fi;
        float acc = 0.f;
        for (int s = 0; s < g.S; ++s) acc = __builtin_fmaf(sA[s * g.S + j], sU[s * g.Fi + k], acc);
        dx[i] = acc;
      }
    }
  }
  float* row = partial + (size_t)blockIdx.x * (npair + g.Fo);
#pragma unroll
  for (int q = 0; q < GA_PAIRS; ++q) {
    const int pr = threadIdx.x + q * GA_THREADS;
    if (pr < npair) row[pr] = accW[q];
  }
  if ((int)threadIdx.x < g.Fo) row[npair + threadIdx.x] = accb;
}

__global__ void gcn_any_reduce_kernel(const float* __restrict__ partial, int rows, int n, int npair, float* __restrict__ dW,
                                      float* __restrict__ db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += partial[(size_t)r * n + i];   // fixed order
  if (i < npair) dW[i] = s;
  else db[i - npair] = s;
}

int ga_grid(int ntiles) { return ntiles < 1024 ? ntiles : 1024; }

// wgnn_gru_fwd / wgnn_gru_bwd hand-over: the caller's dense g [rows][I] <-> the library's padded rows (pitch ld >= I + 1, a
// ones column at I on which b_ih rides, zeros behind it: what gcn32_fwd_kernel writes), and dg [rows][ld] -> [rows][I]
__global__ void pack_g_kernel(const float* __restrict__ gin, size_t rows, int I, float* __restrict__ g, int ld) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * (size_t)ld) return;
  const size_t r = i / ld;
  const int c = (int)(i % ld);
  g[i] = c < I ? gin[r * I + c] : (c == I ? 1.f : 0.f);
}
__global__ void unpack_dg_kernel(const float* __restrict__ dg, size_t rows, int I, int ld, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * (size_t)I) return;
  out[i] = dg[(i / I) * ld + (i % I)];
}

}  // namespace

bool gcn_any_supported(int S, int Fi, int Fo) { return S >= 1 && S <= GA_MAXS && Fi >= 1 && Fi <= GA_MAXF && Fo >= 1 && Fo <= GA_MAXF; }
size_t gcn_any_bwd_partial_floats(int ntiles, int Fi, int Fo) { return (size_t)ga_grid(ntiles) * ((size_t)Fi * Fo + Fo); }

int launch_gcn_any_fwd(int ntiles, int S, int Fi, int Fo, const float* A, const float* X, const float* W, const float* b, float* out,
                       hipStream_t st) {
  const GaGeom g{S, Fi, Fo};
  const size_t smem = sizeof(float) * g.smem_floats(false);
  static std::atomic<unsigned long long> done{0};
  if (ensure_dyn_smem((const void*)gcn_any_fwd_kernel, 160 * 1024, done) != WGNN_OK) return WGNN_ERR_HIP;
  PROF_LAUNCH("gcn_any_fwd_kernel", (double)ntiles * (2.0 * S * S * Fi + 2.0 * S * Fi * Fo), 4.0 * ntiles * S * (Fi + Fo), st,
              hipLaunchKernelGGL(gcn_any_fwd_kernel, dim3(ga_grid(ntiles)), dim3(GA_THREADS), smem, st, g, ntiles, A, X, W, b, out));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_gcn_any_bwd(int ntiles, int S, int Fi, int Fo, const float* A, const float* X, const float* W, const float* out,
                       const float* dout, float* dW, float* db, float* dX, float* partial, hipStream_t st) {
  const GaGeom g{S, Fi, Fo};
  const size_t smem = sizeof(float) * g.smem_floats(true);
  static std::atomic<unsigned long long> done{0};
  if (ensure_dyn_smem((const void*)gcn_any_bwd_kernel, 160 * 1024, done) != WGNN_OK) return WGNN_ERR_HIP;
  const int grid = ga_grid(ntiles), n = Fi * Fo + Fo;
  PROF_LAUNCH("gcn_any_bwd_kernel", (double)ntiles * (2.0 * S * S * Fi * 2 + 2.0 * S * Fi * Fo * 2),
              4.0 * ntiles * S * (Fi + 2 * Fo + (dX ? Fi : 0)), st,
              hipLaunchKernelGGL(gcn_any_bwd_kernel, dim3(grid), dim3(GA_THREADS), smem, st, g, ntiles, A, X, W, out, dout, dX,
                                 partial));
  WGNN_CHECK_LAUNCH();
  PROF_LAUNCH("gcn_any_reduce_kernel", 0.0, 4.0 * grid * n, st,
              hipLaunchKernelGGL(gcn_any_reduce_kernel, dim3(cdiv_i(n, 256)), dim3(256), 0, st, partial, grid, n, Fi * Fo, dW, db));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_pack_g(const float* gin, size_t rows, int I, float* g, int ld, hipStream_t st) {
  const size_t n = rows * (size_t)ld;
  PROF_LAUNCH("pack_g_kernel", 0.0, 4.0 * rows * (I + ld), st,
              hipLaunchKernelGGL(pack_g_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gin, rows, I, g, ld));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
int launch_unpack_dg(const float* dg, size_t rows, int I, int ld, float* out, hipStream_t st) {
  const size_t n = rows * (size_t)I;
  PROF_LAUNCH("unpack_dg_kernel", 0.0, 8.0 * n, st,
              hipLaunchKernelGGL(unpack_dg_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dg, rows, I, ld, out));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
