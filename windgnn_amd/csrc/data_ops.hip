// "Next" rows either side of the hot path:
//   N2  window builder / on-device batcher   src/step4_sequence_preparer.py:7-21
//       x = data[i*L:(i+1)*L, :, 2:15]; y_k = data[i*L+k:(i+1)*L+k, :, 13], k = 1,2,3, concatenated on the
//       station axis.  Here the 13 feature columns are already split off the 2 id columns, so the label
//       column 13 of the reference is feature index 11 ("Wind Speed 10 m Avg.").
//   N4  evaluation read-out                    src/main.py:100-104,116,131,146
//       last timestep of every window, de-normalised: y * (wind_max - wind_min) + wind_min.
// Both are pure HBM-bound gathers (bit-exact copies / one fma).
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) make_windows_kernel(const float* __restrict__ feat, int64_t Ttot, int S, int F,
                                                           int seq, int label_feat, const int32_t* __restrict__ starts,
                                                           int B, float* __restrict__ X, float* __restrict__ L) {
  const int64_t SF = (int64_t)S * F;
  const int64_t nx = (int64_t)B * seq * SF, nl = (int64_t)B * seq * 3 * S;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nx + nl; i += stride) {
    if (i < nx) {
      const int64_t b = i / (seq * SF), r = i % (seq * SF);          // r = t*SF + s*F + f: contiguous copy per window
      const int64_t t0 = starts ? (int64_t)starts[b] : b * seq;
      X[i] = feat[t0 * SF + r];
    } else {
      const int64_t j = i - nx;
      const int64_t b = j / ((int64_t)seq * 3 * S), r = j % ((int64_t)seq * 3 * S);
      const int t = (int)(r / (3 * S)), ks = (int)(r % (3 * S));
      const int k = ks / S, s = ks % S;
      const int64_t t0 = starts ? (int64_t)starts[b] : b * seq;
      L[j] = feat[(t0 + t + k + 1) * SF + (int64_t)s * F + label_feat];
    }
  }
}

__global__ void predict_last_kernel(const float* __restrict__ Y, int B, int T, int H, float wmin, float wmax,
                                    float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i % H;
  out[i] = Y[((size_t)b * T + (T - 1)) * H + j] * (wmax - wmin) + wmin;
}

}  // namespace

extern "C" {

int wgnn_make_windows(const float* feat, int64_t Ttot, int32_t S, int32_t F, int32_t seq_len, int32_t label_feat,
                      const int32_t* starts_host_checked, const int32_t* starts_dev, int32_t B, float* X, float* L,
                      void* stream) {
  if (!feat || !X || !L) return WGNN_ERR_NULL;
  if (Ttot < 1 || S < 1 || F < 1 || seq_len < 1 || B < 1 || label_feat < 0 || label_feat >= F) return WGNN_ERR_SHAPE;
  // every window needs seq_len rows of x plus 3 more rows for the +1/+2/+3 h labels
  if (starts_dev) {
    if (!starts_host_checked) return WGNN_ERR_NULL;      // caller passes the same starts on the host for validation
    for (int b = 0; b < B; ++b)
      if (starts_host_checked[b] < 0 || (int64_t)starts_host_checked[b] + seq_len + 3 > Ttot) return WGNN_ERR_SHAPE;
  } else if ((int64_t)B * seq_len + 3 > Ttot) {
    return WGNN_ERR_SHAPE;
  }
  const int64_t n = (int64_t)B * seq_len * ((int64_t)S * F + 3 * S);
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipStream_t st = (hipStream_t)stream;
  PROF_LAUNCH("make_windows_kernel", 0.0, 8.0 * n, st,
              hipLaunchKernelGGL(make_windows_kernel, dim3(grid), dim3(256), 0, st, feat, Ttot, S, F, seq_len,
                                 label_feat, starts_dev, B, X, L));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int wgnn_predict_last(const float* Y, int32_t B, int32_t T, int32_t H, float wind_min, float wind_max, float* out,
                      void* stream) {
  if (!Y || !out) return WGNN_ERR_NULL;
  if (B < 1 || T < 1 || H < 1) return WGNN_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(predict_last_kernel, dim3(cdiv_i(B * H, 256)), dim3(256), 0, st, Y, B, T, H, wind_min, wind_max,
                     out);
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
}
