// GRU recurrence over the window, forward and backward (BPTT), exact-fp32 MFMA.
//
// Reference: nn.GRU(gru_input, gru_hidden_dim, batch_first=True) called with h0 = 0 at
// src/step6_gcn_gru_combined_model.py:11,23 (torch gate order r,z,n):
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r*gh_n), h = (1-z) n + z h_prev
// with gi = W_ih g + b_ih (precomputed for every timestep by the input-projection GEMM) and
// gh = W_hh h_prev + b_hh computed here.
//
// One workgroup owns 16 windows for all T steps.  W_hh [3H,H] stays resident in LDS for the whole
// launch (124.8 KB at H = 102); h lives in LDS between steps; each wave owns one 16-wide tile of
// hidden units for all three gates, so the gate math is lane-local in the MFMA C layout.
#include "common.h"

namespace {

constexpr int MB = 16;        // windows per workgroup (one MFMA M tile)
constexpr int NTHREADS = 512; // 8 waves -> up to 128 hidden units
constexpr int WPAD = 8;

struct GruGeom {
  int H, KS, HS, NHT, KS3, DS;
  __host__ __device__ explicit GruGeom(int H_) {
    H = H_;
    KS = (H + 3) / 4;
    HS = 4 * KS + 2;
    NHT = (H + 15) / 16;
    KS3 = (3 * H + 3) / 4;
    DS = 4 * KS3 + 2;
  }
  __host__ __device__ size_t fwd_bytes() const { return (size_t)(3 * H * H + WPAD + MB * HS) * 4; }
  __host__ __device__ size_t bwd_bytes() const { return (size_t)(3 * H * H + WPAD + MB * DS) * 4; }
};

__global__ void __launch_bounds__(NTHREADS) gru_fwd_kernel(int B, int T, int H, const float* __restrict__ GI, int ldgi,
                                                           const float* __restrict__ Whh,
                                                           const float* __restrict__ bhh, float* __restrict__ Y,
                                                           float* __restrict__ gates) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GruGeom G(H);
  float* Ws = smem;
  float* hs = Ws + 3 * H * H + WPAD;
  for (int i = threadIdx.x; i < 3 * H * H; i += NTHREADS) Ws[i] = Whh[i];
  for (int i = threadIdx.x; i < WPAD; i += NTHREADS) Ws[3 * H * H + i] = 0.f;
  for (int i = threadIdx.x; i < MB * G.HS; i += NTHREADS) hs[i] = 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int ht = wave;
  const bool active = ht < G.NHT;
  const int j = 16 * ht + lm;
  const bool jv = active && j < H;
  const int jc = jv ? j : H - 1;
  const int b0 = blockIdx.x * MB;
  const float bh_r = bhh[jc], bh_z = bhh[H + jc], bh_n = bhh[2 * H + jc];

  float gi[3][4];
  auto load_gi = [&](int t, float (&dst)[3][4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + 4 * lk + r;
      const bool ok = jv && b < B && t < T;
      const float* row = GI + ((size_t)(ok ? b : 0) * T + (ok ? t : 0)) * ldgi;
      dst[0][r] = ok ? row[j] : 0.f;
      dst[1][r] = ok ? row[H + j] : 0.f;
      dst[2][r] = ok ? row[2 * H + j] : 0.f;
    }
  };
  load_gi(0, gi);

  for (int t = 0; t < T; ++t) {
    float gin[3][4];
    load_gi(t + 1, gin);   // prefetch next step's input projection under this step's MFMAs
    f32x4 ar, az, an;
    float hnew[4];
    if (active) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ar[r] = gi[0][r] + bh_r;
        az[r] = gi[1][r] + bh_z;
        an[r] = bh_n;
      }
      for (int ks = 0; ks < G.KS; ++ks) {
        const int k = 4 * ks + lk;
        const float a = hs[lm * G.HS + k];
        ar = mfma16(a, Ws[jc * H + k], ar);
        az = mfma16(a, Ws[(H + jc) * H + k], az);
        an = mfma16(a, Ws[(2 * H + jc) * H + k], an);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * lk + r;
        // hardware exp2 / rcp forms (|error| < 3e-7, as in grux.hip and gru_small.hip): 12 VALU per element instead of ~90
        const float rg = sigmoid_fast(ar[r]);
        const float zg = sigmoid_fast(az[r]);
        const float ng = tanh_fast(gi[2][r] + rg * an[r]);
        const float hold = hs[m * G.HS + jc];
        hnew[r] = (1.f - zg) * ng + zg * hold;
        const int b = b0 + m;
        if (jv && b < B) {
          const size_t bt = (size_t)b * T + t;
          Y[bt * H + j] = hnew[r];
          if (gates)   // one 16-byte record (r, z, n, gh_n) per element: [bt][j][4], read back as one load by gru_bwd
            *(f32x4*)(gates + (bt * H + j) * 4) = f32x4{rg, zg, ng, an[r]};
        }
      }
    }
    __syncthreads();
    if (jv) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hs[(4 * lk + r) * G.HS + j] = hnew[r];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) gi[g][r] = gin[g][r];
  }
}

// BPTT.  Per step (t descending), with dh = dY_t + dh_next:
//   dn = dh (1-z), dz = dh (hprev - n), dnt = dn (1-n^2), dr = dnt gh_n,
//   dar = dr r (1-r), daz = dz z (1-z);  dgi = [dar, daz, dnt], dgh = [dar, daz, dnt r]
//   dh_next = dh z + dgh W_hh
// dGI/dGH rows are written for the weight-gradient GEMMs that follow.
__global__ void __launch_bounds__(NTHREADS) gru_bwd_kernel(int B, int T, int H, const float* __restrict__ Whh,
                                                           const float* __restrict__ Y, const float* __restrict__ dY,
                                                           const float* __restrict__ gates, float* __restrict__ dGI,
                                                           float* __restrict__ dGH, int ldd) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GruGeom G(H);
  float* Ws = smem;
  float* ds = Ws + 3 * H * H + WPAD;
  for (int i = threadIdx.x; i < 3 * H * H; i += NTHREADS) Ws[i] = Whh[i];
  for (int i = threadIdx.x; i < WPAD; i += NTHREADS) Ws[3 * H * H + i] = 0.f;
  for (int i = threadIdx.x; i < MB * G.DS; i += NTHREADS) ds[i] = 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int ht = wave;
  const bool active = ht < G.NHT;
  const int j = 16 * ht + lm;
  const bool jv = active && j < H;
  const int jc = jv ? j : H - 1;
  const int b0 = blockIdx.x * MB;
  const int G3 = 3 * H;
  {  // zero the K-padding columns [3H, ldd) of this workgroup's dGI/dGH rows
    const int npad = ldd - G3;
    const int nrows = min(MB, B - b0) * T;
    for (int i = threadIdx.x; i < nrows * npad; i += NTHREADS) {
      const size_t o = ((size_t)b0 * T + i / npad) * ldd + G3 + i % npad;
      dGI[o] = 0.f;
      dGH[o] = 0.f;
    }
  }

  struct StepIn { float dy[4], r[4], z[4], n[4], ghn[4], hp[4]; };
  auto load_step = [&](int t, StepIn& s) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + 4 * lk + r;
      const bool ok = jv && b < B && t >= 0;
      const size_t bt = (size_t)(ok ? b : 0) * T + (ok ? t : 0);
      const f32x4 gq = ok ? *(const f32x4*)(gates + (bt * H + j) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};   // gru_fwd's record
      s.dy[r] = ok ? dY[bt * H + j] : 0.f;
      s.r[r] = gq[0];
      s.z[r] = gq[1];
      s.n[r] = gq[2];
      s.ghn[r] = gq[3];
      s.hp[r] = (ok && t > 0) ? Y[(bt - 1) * H + j] : 0.f;
    }
  };
  StepIn cur;
  load_step(T - 1, cur);
  f32x4 dhn = {0.f, 0.f, 0.f, 0.f};

  for (int t = T - 1; t >= 0; --t) {
    StepIn nxt;
    load_step(t - 1, nxt);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * lk + r;
      const int b = b0 + m;
      const float dh = cur.dy[r] + dhn[r];
      const float rg = cur.r[r], zg = cur.z[r], ng = cur.n[r];
      const float dn = dh * (1.f - zg);
      const float dz = dh * (cur.hp[r] - ng);
      const float dnt = dn * (1.f - ng * ng);
      const float dr = dnt * cur.ghn[r];
      const float dar = dr * rg * (1.f - rg);
      const float daz = dz * zg * (1.f - zg);
      const float dnr = dnt * rg;
      acc[r] = dh * zg;
      if (jv) {
        ds[m * G.DS + j] = dar;
        ds[m * G.DS + H + j] = daz;
        ds[m * G.DS + 2 * H + j] = dnr;
        if (b < B) {
          const size_t bt = (size_t)b * T + t;
          float* gi = dGI + bt * ldd;
          float* gh = dGH + bt * ldd;
          gi[j] = dar; gi[H + j] = daz; gi[2 * H + j] = dnt;
          gh[j] = dar; gh[H + j] = daz; gh[2 * H + j] = dnr;
        }
      }
    }
    __syncthreads();
    if (active && t > 0) {
      // dgh W_hh over k = 3H: four independent accumulator chains (one 77-long dependent chain of 16x16x4 MFMAs was
      // 2800 cycles of pure latency per step)
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
      int ks = 0;
      for (; ks + 4 <= G.KS3; ks += 4) {
        const int k = 4 * ks + lk;             // k + 12 < 4 KS3: only the last k step can pass 3H
        acc = mfma16(ds[lm * G.DS + k], Ws[min(k, G3 - 1) * H + jc], acc);
        a1 = mfma16(ds[lm * G.DS + k + 4], Ws[min(k + 4, G3 - 1) * H + jc], a1);
        a2 = mfma16(ds[lm * G.DS + k + 8], Ws[min(k + 8, G3 - 1) * H + jc], a2);
        a3 = mfma16(ds[lm * G.DS + k + 12], Ws[min(k + 12, G3 - 1) * H + jc], a3);
      }
      for (; ks < G.KS3; ++ks) {
        const int k = 4 * ks + lk;
        const int kc = k < G3 ? k : G3 - 1;   // ds[., k >= 3H] is zero, so the clamped weight is harmless
        acc = mfma16(ds[lm * G.DS + k], Ws[kc * H + jc], acc);
      }
      acc = (acc + a1) + (a2 + a3);
    }
    dhn = acc;
    __syncthreads();
    cur = nxt;
  }
}

}  // namespace

bool gru_shape_supported(int H) {
  if (H < 1 || H > 16 * (NTHREADS / 64)) return false;
  GruGeom G(H);
  return G.bwd_bytes() <= 160 * 1024 && G.fwd_bytes() <= 160 * 1024;
}

int launch_gru_fwd(int B, int T, int H, const float* GI, int ldgi, const float* Whh, const float* bhh, float* Y,
                   float* gates, hipStream_t st) {
  GruGeom G(H);
  size_t smem = G.fwd_bytes();
  static std::atomic<unsigned long long> done{0};   // smem depends on H: re-arm if it grows
  static std::atomic<size_t> armed{0};
  if (smem > armed.load()) { done.store(0); armed.store(smem); }
  if (ensure_dyn_smem((const void*)gru_fwd_kernel, armed.load(), done) != WGNN_OK) return WGNN_ERR_HIP;
  const double bt = (double)B * T;
  PROF_LAUNCH("gru_fwd_kernel", bt * 2.0 * 3 * H * H, bt * 4.0 * (3 * H + H + (gates ? 4 * H : 0)), st,
              hipLaunchKernelGGL(gru_fwd_kernel, dim3(cdiv_i(B, MB)), dim3(NTHREADS), smem, st, B, T, H, GI, ldgi, Whh,
                                 bhh, Y, gates));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int launch_gru_bwd(int B, int T, int H, const float* Whh, const float* Y, const float* dY, const float* gates,
                   float* dGI, float* dGH, int ldd, hipStream_t st) {
  GruGeom G(H);
  size_t smem = G.bwd_bytes();
  static std::atomic<unsigned long long> done{0};
  static std::atomic<size_t> armed{0};
  if (smem > armed.load()) { done.store(0); armed.store(smem); }
  if (ensure_dyn_smem((const void*)gru_bwd_kernel, armed.load(), done) != WGNN_OK) return WGNN_ERR_HIP;
  const double bt = (double)B * T;
  PROF_LAUNCH("gru_bwd_kernel", bt * 2.0 * 3 * H * H, bt * 4.0 * (4 * H + 2 * H + 6 * H), st,
              hipLaunchKernelGGL(gru_bwd_kernel, dim3(cdiv_i(B, MB)), dim3(NTHREADS), smem, st, B, T, H, Whh, Y, dY,
                                 gates, dGI, dGH, ldd));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
