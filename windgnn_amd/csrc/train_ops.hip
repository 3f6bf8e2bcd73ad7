// Call-site ops of the reference training step that sit either side of the hot path:
//   nn.MSELoss()(outputs, batch_y)      src/main.py:49,72   -> loss and dY = 2 (Y-L) scale / n
//   torch.optim.Adam(lr=1e-3).step()    src/main.py:52,80   -> flat-buffer Adam
//   amax_scale: power-of-two range scale of the f16x3 backward, scales = {2^k, 2^-k} with 2^k*max|dY| in [1,2)
// All are HBM-bound elementwise passes; float4 where alignment allows.
#include "common.h"

namespace {

constexpr int MSE_BLOCKS = 1024;   // partials fit the 4096-byte workspace the ABI asks for

__global__ void __launch_bounds__(256) mse_kernel(const float* __restrict__ Y, const float* __restrict__ L,
                                                  int64_t n, float gscale, float* __restrict__ dY,
                                                  float* __restrict__ part) {
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool al = ((((uintptr_t)Y) | ((uintptr_t)L) | ((uintptr_t)dY)) & 15) == 0;
  const int64_t n4 = al ? n / 4 : 0;
  const f32x4* Y4 = (const f32x4*)Y;
  const f32x4* L4 = (const f32x4*)L;
  f32x4* D4 = (f32x4*)dY;
  for (int64_t i = gid; i < n4; i += stride) {
    const f32x4 d = Y4[i] - L4[i];
    s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    if (dY) D4[i] = d * gscale;
  }
  for (int64_t i = 4 * n4 + gid; i < n; i += stride) {
    const float d = Y[i] - L[i];
    s += d * d;
    if (dY) dY[i] = d * gscale;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void mse_finalize_kernel(const float* __restrict__ part, int nblk, float inv_n, float* loss) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 64) s += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (threadIdx.x == 0) loss[0] = s * inv_n;
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr_over_bc1, float inv_sqrt_bc2, float b1, float b2,
                                                   float eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] -= lr_over_bc1 * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
}

// scales[0] = 2^-floor(log2(max|x|)) (1 if the max is 0 or not finite), scales[1] = 1/scales[0]
__global__ void __launch_bounds__(256) amax_partial_kernel(const float* __restrict__ x, int64_t n,
                                                           float* __restrict__ part) {
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = (((uintptr_t)x & 15) == 0) ? n / 4 : 0;
  const f32x4* x4 = (const f32x4*)x;
  for (int64_t i = gid; i < n4; i += stride) {
    const f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  for (int64_t i = 4 * n4 + gid; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(ws[0], ws[1]), fmaxf(ws[2], ws[3]));
}

__global__ void amax_finalize_kernel(const float* __restrict__ part, int nblk, float* scales) {
  float m = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 64) m = fmaxf(m, part[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if (threadIdx.x == 0) {
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)
      s = ldexpf(1.f, 1 - e);        // s*m in [1, 2)
    }
    scales[0] = s;
    scales[1] = 1.f / s;
  }
}

// Fused-loss backward (wgnn_bwd_mse_part): the loss and the range scale of dY = 2 (Y - L) grad_scale / n in ONE
// pass over Y and L, without writing dY.  part[b] = sum (Y-L)^2, part[nblk + b] = max |Y-L| of block b.
__global__ void __launch_bounds__(256) mse_stats_kernel(const float* __restrict__ Y, const float* __restrict__ L,
                                                        int64_t n, float* __restrict__ part) {
  __shared__ float red[256], redm[256];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = ((((uintptr_t)Y) | ((uintptr_t)L)) & 15) == 0 ? n / 4 : 0;
  float acc = 0.f, m = 0.f;
  int64_t i = gid;
  for (; i + 3 * stride < n4; i += 4 * stride) {          // four independent 16-byte load pairs in flight
    f32x4 d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = ((const f32x4*)Y)[i + u * stride] - ((const f32x4*)L)[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc += d[u][0] * d[u][0] + d[u][1] * d[u][1] + d[u][2] * d[u][2] + d[u][3] * d[u][3];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(d[u][0]), fabsf(d[u][1]))), fmaxf(fabsf(d[u][2]), fabsf(d[u][3])));
    }
  }
  for (; i < n4; i += stride) {
    const f32x4 d = ((const f32x4*)Y)[i] - ((const f32x4*)L)[i];
    acc += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
  }
  for (int64_t k = 4 * n4 + gid; k < n; k += stride) {
    const float d = Y[k] - L[k];
    acc += d * d;
    m = fmaxf(m, fabsf(d));
  }
  red[threadIdx.x] = acc;
  redm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[threadIdx.x] += red[threadIdx.x + s];
      redm[threadIdx.x] = fmaxf(redm[threadIdx.x], redm[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x] = red[0];
    part[gridDim.x + blockIdx.x] = redm[0];
  }
}

// loss = sum / n; scales = {2^k, 2^-k, coef} with coef = 2 grad_scale / n and 2^k * coef * max|Y-L| in [1, 2)
__global__ void mse_stats_finalize_kernel(const float* __restrict__ part, int nblk, float inv_n, float coef,
                                          float* __restrict__ loss, float* __restrict__ scales) {
  float s = 0.f, m = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 64) {
    s += part[i];
    m = fmaxf(m, part[nblk + i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    m = fmaxf(m, __shfl_xor(m, o, 64));
  }
  if (threadIdx.x == 0) {
    loss[0] = s * inv_n;
    m *= fabsf(coef);
    float sc = 1.f;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);
      sc = ldexpf(1.f, 1 - e);
    }
    scales[0] = sc;
    scales[1] = 1.f / sc;
    scales[2] = coef;
  }
}

}  // namespace

// loss and scales {2^k, 2^-k, 2 grad_scale / n} of the fused-loss backward; part: >= 2048 floats
int launch_mse_stats(const float* Y, const float* L, int64_t n, float grad_scale, float* loss, float* scales,
                     float* part, hipStream_t st) {
  PROF_LAUNCH("mse_stats_kernel", 4.0 * n, 8.0 * n, st,
              hipLaunchKernelGGL(mse_stats_kernel, dim3(MSE_BLOCKS), dim3(256), 0, st, Y, L, n, part));
  WGNN_CHECK_LAUNCH();
  hipLaunchKernelGGL(mse_stats_finalize_kernel, dim3(1), dim3(64), 0, st, part, MSE_BLOCKS, 1.0f / (float)n,
                     2.0f * grad_scale / (float)n, loss, scales);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int mse_stats_blocks() { return MSE_BLOCKS; }

int launch_amax_scale(const float* x, int64_t n, float* scales, float* part /*>=448 floats*/, hipStream_t st) {
  PROF_LAUNCH("amax_partial_kernel", (double)n, 4.0 * n, st,
              hipLaunchKernelGGL(amax_partial_kernel, dim3(448), dim3(256), 0, st, x, n, part));
  WGNN_CHECK_LAUNCH();
  hipLaunchKernelGGL(amax_finalize_kernel, dim3(1), dim3(64), 0, st, part, 448, scales);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_mse(const float* Y, const float* L, int64_t n, float scale, float* dY, float* loss, float* ws,
               hipStream_t st) {
  PROF_LAUNCH("mse_kernel", 3.0 * n, 12.0 * n, st,
              hipLaunchKernelGGL(mse_kernel, dim3(MSE_BLOCKS), dim3(256), 0, st, Y, L, n, 2.0f * scale / (float)n, dY,
                                 ws));
  WGNN_CHECK_LAUNCH();
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(64), 0, st, ws, MSE_BLOCKS, 1.0f / (float)n, loss);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1, float b2,
                float eps, hipStream_t st) {
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  PROF_LAUNCH("adam_kernel", 10.0 * n, 28.0 * n, st,
              hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n,
                                 (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), b1, b2, eps));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
