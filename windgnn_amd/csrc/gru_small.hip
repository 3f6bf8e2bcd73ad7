// GRU recurrence for SMALL batches, exact fp32: one to four windows per workgroup, W_hh in REGISTERS.
//
// Reference: nn.GRU (1 layer, batch_first, h0 = 0, gates r,z,n) at src/step6_gcn_gru_combined_model.py:11,23 and
// its BPTT (src/main.py:79) -- the same equations as gru.hip.
//
// gru.hip gives a workgroup 16 windows (one MFMA M tile): at BASELINE configs[1] (B = 256) that is 16 workgroups on
// 256 CUs, and the launch lasts as long as ONE workgroup's 24 dependent steps (126 / 163 us forward / backward).
// Here a workgroup owns ONE window (the WPB template parameter stays 1: more windows per workgroup spill), thread i owns gate row i of W_hh (forward: the row itself, 4H bytes of
// registers; backward: column j of gate block q, i = q H + j) and the per-step matrix-vector products are plain fp32
// FMA chains against h / dgh broadcast from LDS: no MFMA tile to fill, ~B workgroups instead of B/16, and a step costs
// ~H FMAs per thread.  Used when B <= 768; results are fp32 fmaf chains like gru.hip's (different summation order).
// Round 5: the contraction of a row is split over KSP threads (KSP parts of the workgroup, part p owns k in [p KC, (p + 1) KC),
// KC = HMAX / KSP; 3 parts at H = 102: 960 threads instead of 320), the parts' sums meet in LDS and the gate phase adds them in
// a fixed order.  A step was ~2000 cycles of which ~1000 were the 102-long FMA chains of two waves per SIMD (the reference's
// own call shape, ONE window of 168 steps, spends 73 % of its step in these two kernels: profiles/r5_dropin_latency.txt).
#include "common.h"

namespace {

constexpr int SMALL_THREADS = 1024;  // most threads a workgroup may have: KSP parts of roundup(3 H, 64) threads each
// Instances: HMAX 32 / 64 / 96 / 108 (H <= 106: three parts of 320 threads) / 112 / 128; parts per instance:
constexpr int small_ksp(int HMAX) { return HMAX <= 32 ? 1 : ((HMAX == 96 || HMAX == 108) ? 3 : 2); }
// launch bound of an instance = its largest workgroup (the register budget follows from it: 768 threads = 3 waves per SIMD)
constexpr int small_threads(int HMAX) {
  return small_ksp(HMAX) * ((3 * HMAX + 63) / 64 * 64) < SMALL_THREADS ? small_ksp(HMAX) * ((3 * HMAX + 63) / 64 * 64) : SMALL_THREADS;
}
static int small_hmax(int H) { return H <= 32 ? 32 : H <= 64 ? 64 : H <= 96 ? 96 : H <= 106 ? 108 : H <= 112 ? 112 : 128; }

// ------------------------------------------------------------------------------------------------
template <int HMAX, int WPB, int KSP>
__global__ void __launch_bounds__(small_threads(HMAX)) gru_small_fwd_kernel(int B, int T, int H,
                                                                      const float* __restrict__ GI, int ldgi,
                                                                      const float* __restrict__ Whh,
                                                                      const float* __restrict__ bhh,
                                                                      float* __restrict__ Y, float* __restrict__ gates,
                                                                      int stage_w, float* __restrict__ hprev, int hq,
                                                                      const float* __restrict__ Lab,
                                                                      float* __restrict__ stat_part) {
  // Lab + stat_part (wgnn_fwd_loss): this workgroup's sum of (h - label)^2 (and max |h - label|) goes to
  // stat_part[blockIdx.x] / stat_part[gridDim.x + blockIdx.x], the tag behind them says so; a forward without labels
  // clears the tag (as gru.hip / grux.hip)
  if (stat_part && blockIdx.x == 0 && threadIdx.x == 0) stat_part[2 * gridDim.x] = Lab ? WGNN_STATS_TAG : 0.f;
  float ssum = 0.f, smax = 0.f;
  // hprev (nullable): rows [h_{t-1} | 1 | 0..] of width hq with 16-byte aligned rows, the B operand of the dW_hh GEMM
  // (gemm32.hip) -- written here instead of by a pass of its own over Y
  if (hprev) {
    for (int wdw = 0; wdw < WPB; ++wdw) {
      const int b = blockIdx.x * WPB + wdw;
      if (b >= B) break;
      for (int q = threadIdx.x; q < T * (hq - H); q += blockDim.x) {
        const int c = H + q % (hq - H);
        hprev[((size_t)b * T + q / (hq - H)) * hq + c] = c == H ? 1.f : 0.f;
      }
      for (int c = threadIdx.x; c < H; c += blockDim.x) hprev[(size_t)b * T * hq + c] = 0.f;
    }
  }
  constexpr int KC = HMAX / KSP;                                     // k extent of one part
  static_assert(HMAX % KSP == 0 && KC % 4 == 0, "parts of whole 16-byte groups");
  __shared__ __attribute__((aligned(16))) float hs[WPB][HMAX];       // h_{t-1}, zero beyond H
  __shared__ float ghs[KSP][WPB][3 * HMAX];                          // the parts' shares of W_hh h (+ b_hh in part 0), gate-major with stride HMAX
  const int TP = (int)blockDim.x / KSP;                              // threads per part = roundup(3 H, 64)
  const int part_id = (int)threadIdx.x / TP, i = (int)threadIdx.x % TP;
  const int k0 = part_id * KC;
  const int G3 = 3 * H;
  const bool iv = i < G3;
  const int ic = iv ? i : G3 - 1;
  const int q = ic / H, jq = ic % H;                                  // gate and unit of this thread's row
  // This thread's row of W_hh into registers.  Read straight from HBM / L2 each lane of a load would touch its own
  // cache line (rows are H floats apart): 112 loads x 64 lines per wave, ~33 us per workgroup.  So W_hh comes in
  // linearly (coalesced) through LDS when it fits (3 H^2 floats of dynamic LDS: H <= 108) and each thread picks its
  // row out of the staged copy; larger H takes the strided loads.
  float w[KC];                                                        // this thread's KC elements of its row: k0 .. k0 + KC - 1
  extern __shared__ __attribute__((aligned(16))) float wst[];
  if (stage_w) {
    const int nw = G3 * H;
    if ((nw & 3) == 0 && (((uintptr_t)Whh) & 15) == 0) {    // 16-byte copies, several in flight per thread
      for (int e = threadIdx.x; e < nw / 4; e += blockDim.x) ((f32x4*)wst)[e] = ((const f32x4*)Whh)[e];
    } else {
      for (int e = threadIdx.x; e < nw; e += blockDim.x) wst[e] = Whh[e];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KC; ++k) w[k] = k0 + k < H ? wst[ic * H + k0 + k] : 0.f;
  } else {
#pragma unroll
    for (int k = 0; k < KC; ++k) w[k] = k0 + k < H ? Whh[(size_t)ic * H + k0 + k] : 0.f;
  }
  const float bh = part_id == 0 ? bhh[ic] : 0.f;
  const int b0 = blockIdx.x * WPB;
  for (int k = threadIdx.x; k < WPB * HMAX; k += blockDim.x) (&hs[0][0])[k] = 0.f;
  // gate phase: thread j < H owns hidden unit j of every window of the workgroup
  const int j = threadIdx.x;
  const bool jv = j < H;
  const int jc = jv ? j : H - 1;
  float gi[WPB][4], gin[WPB][4];                                      // [3] = the label of the step (fused loss)
  auto load_gi = [&](int t, float (&dst)[WPB][4]) {
    const int tc = t < T ? t : T - 1;
#pragma unroll
    for (int wdw = 0; wdw < WPB; ++wdw) {
      const int b = b0 + wdw < B ? b0 + wdw : B - 1;
      const float* row = GI + ((size_t)b * T + tc) * ldgi;
      dst[wdw][0] = row[jc];
      dst[wdw][1] = row[H + jc];
      dst[wdw][2] = row[2 * H + jc];
      dst[wdw][3] = Lab ? Lab[((size_t)b * T + tc) * H + jc] : 0.f;
    }
  };
  load_gi(0, gi);
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    load_gi(t + 1, gin);                            // next step's input projection, in flight under this step
    float acc[WPB];
#pragma unroll
    for (int wdw = 0; wdw < WPB; ++wdw) {
      // four independent partial sums (k = 0,1,2,3 mod 4): the FMA chain is 4x shorter than H; fixed order => deterministic
      float a0 = bh, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < KC / 4; ++k4) {
        const f32x4 hv = *(const f32x4*)&hs[wdw][k0 + 4 * k4];     // same address in every lane of a part: LDS broadcast
        a0 = fmaf(w[4 * k4 + 0], hv[0], a0);
        a1 = fmaf(w[4 * k4 + 1], hv[1], a1);
        a2 = fmaf(w[4 * k4 + 2], hv[2], a2);
        a3 = fmaf(w[4 * k4 + 3], hv[3], a3);

      }
      acc[wdw] = (a0 + a1) + (a2 + a3);
    }
    if (iv) {
#pragma unroll
      for (int wdw = 0; wdw < WPB; ++wdw) ghs[part_id][wdw][q * HMAX + jq] = acc[wdw];
    }
    __syncthreads();                                // gh complete; every read of h_{t-1} is done
    if (jv) {
#pragma unroll
      for (int wdw = 0; wdw < WPB; ++wdw) {
        const int b = b0 + wdw;
        float ghr = ghs[0][wdw][j], ghz = ghs[0][wdw][HMAX + j], ghn = ghs[0][wdw][2 * HMAX + j];
#pragma unroll
        for (int pp = 1; pp < KSP; ++pp) {                             // the parts' shares, in a fixed order
          ghr += ghs[pp][wdw][j];
          ghz += ghs[pp][wdw][HMAX + j];
          ghn += ghs[pp][wdw][2 * HMAX + j];
        }
        const float rg = sigmoid_fast(gi[wdw][0] + ghr);              // v_exp / v_rcp forms, |error| < 3e-7 (common.h)
        const float zg = sigmoid_fast(gi[wdw][1] + ghz);
        const float ng = tanh_fast(gi[wdw][2] + rg * ghn);
        const float hnew = (1.f - zg) * ng + zg * hs[wdw][j];
        hs[wdw][j] = hnew;
        if (b < B) {
          const size_t bt = (size_t)b * T + t;
          Y[bt * H + j] = hnew;
          if (hprev && t + 1 < T) hprev[(bt + 1) * hq + j] = hnew;
          if (Lab) {
            const float dl = hnew - gi[wdw][3];
            ssum = fmaf(dl, dl, ssum);
            smax = fmaxf(smax, fabsf(dl));
          }
          if (gates)   // one 16-byte record (r, z, n, gh_n) per element, [bt][j][4]
            *(f32x4*)(gates + (bt * H + j) * 4) = f32x4{rg, zg, ng, ghn};
        }
      }
    }
    __syncthreads();                                // h_t in place
#pragma unroll
    for (int wdw = 0; wdw < WPB; ++wdw)
#pragma unroll
      for (int c = 0; c < 4; ++c) gi[wdw][c] = gin[wdw][c];
  }
  if (Lab) {   // fixed order: lanes (xor tree), then the waves
    __shared__ float red[2][SMALL_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ssum += __shfl_xor(ssum, o, 64);
      smax = fmaxf(smax, __shfl_xor(smax, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = ssum;
      red[1][threadIdx.x >> 6] = smax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = 0.f, m = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        a += red[0][w];
        m = fmaxf(m, red[1][w]);
      }
      stat_part[blockIdx.x] = a;
      stat_part[gridDim.x + blockIdx.x] = m;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// BPTT.  Per step (t descending), with dh = dY_t + dh_next:
//   dn = dh (1-z), dz = dh (hprev - n), dnt = dn (1-n^2), dr = dnt gh_n,
//   dar = dr r (1-r), daz = dz z (1-z);  dgi = [dar, daz, dnt], dgh = [dar, daz, dnt r]
//   dh_next = dh z + dgh W_hh                       (thread (q, j): sum over gate block q of dgh[q][k] W_hh[qH+k][j])
template <int HMAX, int WPB, int KSP>
__global__ void __launch_bounds__(small_threads(HMAX)) gru_small_bwd_kernel(int B, int T, int H,
                                                                      const float* __restrict__ Whh,
                                                                      const float* __restrict__ Y,
                                                                      const float* __restrict__ dY,
                                                                      const float* __restrict__ gates,
                                                                      float* __restrict__ dGI, float* __restrict__ dGH,
                                                                      int ldd, const float* __restrict__ Lab,
                                                                      float coef_lab, const float* __restrict__ stat_part,
                                                                      int nstat, float inv_n, float* __restrict__ loss_out,
                                                                      unsigned* status) {
  // Lab: dY = (Y - Lab) * coef_lab is formed here (Y[b, t] is the h_prev this kernel loads for step t + 1 anyway);
  // stat_part: workgroup 0 finalises loss = (sum of the forward's nstat partial sums) / n in a fixed order
  if (stat_part && blockIdx.x == 0) {
    __shared__ float sred[SMALL_THREADS / 64];
    float a = 0.f;
    for (int e = threadIdx.x; e < nstat; e += blockDim.x) a += stat_part[e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      a = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += sred[w];
      const bool tagged = stat_part[2 * nstat] == WGNN_STATS_TAG;
      loss_out[0] = tagged ? a * inv_n : __builtin_nanf("");
      if (!tagged && status) atomicOr(status, WGNN_STATUS_NO_LOSS_STATS);
    }
  }
  constexpr int KC = HMAX / KSP;
  static_assert(HMAX % KSP == 0 && KC % 4 == 0, "parts of whole 16-byte groups");
  __shared__ __attribute__((aligned(16))) float dghs[WPB][3 * HMAX];  // dgh_t, gate-major with stride HMAX, zero pads
  __shared__ float part[KSP][WPB][3][HMAX];                            // the (k part, gate block) shares of dgh W_hh
  const int TP = (int)blockDim.x / KSP;
  const int part_id = (int)threadIdx.x / TP, ip = (int)threadIdx.x % TP;
  const int k0 = part_id * KC;
  const int G3 = 3 * H;
  const int i = part_id == 0 ? ip : 1 << 20;                           // gate-phase index: only part 0's threads own units
  const bool iv = ip < G3;
  const int ic = iv ? ip : G3 - 1;
  const int q = ic / H, j = ic % H;
  float wT[KC];                                                        // W_hh[qH + k0 + k][j]
#pragma unroll
  for (int k = 0; k < KC; ++k) wT[k] = k0 + k < H ? Whh[(size_t)(q * H + k0 + k) * H + j] : 0.f;
  const int b0 = blockIdx.x * WPB;
  for (int k = threadIdx.x; k < WPB * 3 * HMAX; k += blockDim.x) (&dghs[0][0])[k] = 0.f;
  {  // zero the K-padding columns [3H, ldd) of this workgroup's dGI / dGH rows
    const int npad = ldd - G3;
    const int nrows = min(WPB, B - b0) * T;
    for (int e = threadIdx.x; e < nrows * npad; e += blockDim.x) {
      const size_t o = ((size_t)b0 * T + e / npad) * ldd + G3 + e % npad;
      dGI[o] = 0.f;
      dGH[o] = 0.f;
    }
  }
  const bool own = i < H;                                             // gate-phase owner of unit i (then q == 0, j == i)
  struct StepIn { float dy, r, z, n, ghn, hp; };
  StepIn cur[WPB], nxt[WPB];
  auto load_step = [&](int t, StepIn (&s)[WPB]) {
    const int tc = t > 0 ? t : 0;
    if (!own) return;                               // only the H gate-phase owners need a step's inputs
#pragma unroll
    for (int wdw = 0; wdw < WPB; ++wdw) {
      const int b = b0 + wdw < B ? b0 + wdw : B - 1;
      const size_t bt = (size_t)b * T + tc;
      const int jj = i;
      const f32x4 gq = *(const f32x4*)(gates + (bt * H + jj) * 4);   // the forward's (r, z, n, gh_n) record
      s[wdw].dy = Lab ? Lab[bt * H + jj] : dY[bt * H + jj];          // the label, or dY itself
      s[wdw].r = gq[0];
      s[wdw].z = gq[1];
      s[wdw].n = gq[2];
      s[wdw].ghn = gq[3];
      const float hp = Y[(bt - (tc > 0 ? 1 : 0)) * H + jj];
      s[wdw].hp = tc > 0 ? hp : 0.f;
    }
  };
  load_step(T - 1, cur);
  float dhn[WPB], dhz[WPB], ycur[WPB];                                // ycur = Y[b, t]
#pragma unroll
  for (int wdw = 0; wdw < WPB; ++wdw) {
    dhn[wdw] = 0.f;
    const int b = b0 + wdw < B ? b0 + wdw : B - 1;
    ycur[wdw] = (Lab && own) ? Y[((size_t)b * T + T - 1) * H + i] : 0.f;
  }
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    load_step(t - 1, nxt);
    if (own) {
#pragma unroll
      for (int wdw = 0; wdw < WPB; ++wdw) {
        const int b = b0 + wdw;
        const float dh = (Lab ? (ycur[wdw] - cur[wdw].dy) * coef_lab : cur[wdw].dy) + dhn[wdw];
        const float rg = cur[wdw].r, zg = cur[wdw].z, ng = cur[wdw].n;
        const float dn = dh * (1.f - zg);
        const float dz = dh * (cur[wdw].hp - ng);
        const float dnt = dn * (1.f - ng * ng);
        const float dr = dnt * cur[wdw].ghn;
        const float dar = dr * rg * (1.f - rg);
        const float daz = dz * zg * (1.f - zg);
        const float dnr = dnt * rg;
        dhz[wdw] = dh * zg;
        const bool ok = b < B;
        dghs[wdw][i] = ok ? dar : 0.f;
        dghs[wdw][HMAX + i] = ok ? daz : 0.f;
        dghs[wdw][2 * HMAX + i] = ok ? dnr : 0.f;
        if (ok) {
          const size_t bt = (size_t)b * T + t;
          float* gi = dGI + bt * ldd;
          float* gh = dGH + bt * ldd;
          gi[i] = dar; gi[H + i] = daz; gi[2 * H + i] = dnt;
          gh[i] = dar; gh[H + i] = daz; gh[2 * H + i] = dnr;
        }
      }
    }
    __syncthreads();
    if (t > 0) {
      float p[WPB];
#pragma unroll
      for (int wdw = 0; wdw < WPB; ++wdw) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < KC / 4; ++k4) {
          const f32x4 dv = *(const f32x4*)&dghs[wdw][q * HMAX + k0 + 4 * k4];
          a0 = fmaf(dv[0], wT[4 * k4 + 0], a0);
          a1 = fmaf(dv[1], wT[4 * k4 + 1], a1);
          a2 = fmaf(dv[2], wT[4 * k4 + 2], a2);
          a3 = fmaf(dv[3], wT[4 * k4 + 3], a3);

        }
        p[wdw] = (a0 + a1) + (a2 + a3);
      }
      if (iv) {
#pragma unroll
        for (int wdw = 0; wdw < WPB; ++wdw) part[part_id][wdw][q][j] = p[wdw];
      }
    }
    __syncthreads();
    if (own && t > 0) {
#pragma unroll
      for (int wdw = 0; wdw < WPB; ++wdw)
      {
        float acc = 0.f;
#pragma unroll
        for (int pp = 0; pp < KSP; ++pp) acc += (part[pp][wdw][0][i] + part[pp][wdw][1][i]) + part[pp][wdw][2][i];   // fixed order
        dhn[wdw] = dhz[wdw] + acc;
      }
    }
#pragma unroll
    for (int wdw = 0; wdw < WPB; ++wdw) {
      ycur[wdw] = cur[wdw].hp;                                        // Y[b, t - 1]
      cur[wdw] = nxt[wdw];
    }
  }
}

int pick_wpb(int B) { return 1; }   // 2 and 4 windows per workgroup spill at H > 96 (604-848 B of scratch): 409 us at B = 1024

}  // namespace

// B workgroups of ~25-40 us each on 256 CUs: 38 + 39 us at B = 256, 128 + 151 us at B = 1024, against gru.hip's B/16
// workgroups of 113 + 117 us whatever B <= 4096 is (round 3: register-resident W_hh): the one-window form wins up to ~768
bool gru_small_supported(int B, int H) { return H >= 1 && H <= 128 && B <= 768; }

#define SMALL_GO(KERNEL, HM, ...) \
  hipLaunchKernelGGL((KERNEL<HM, 1, small_ksp(HM)>), grid, dim3(small_ksp(HM) * cdiv_i(3 * H, 64) * 64), 0, st, __VA_ARGS__)
#define SMALL_DISPATCH(KERNEL, ...)                                                                              \
  do {                                                                                                           \
    const dim3 grid(B);                                                                                          \
    switch (small_hmax(H)) {                                                                                     \
      case 32: SMALL_GO(KERNEL, 32, __VA_ARGS__); break;                                                         \
      case 64: SMALL_GO(KERNEL, 64, __VA_ARGS__); break;                                                         \
      case 96: SMALL_GO(KERNEL, 96, __VA_ARGS__); break;                                                         \
      case 108: SMALL_GO(KERNEL, 108, __VA_ARGS__); break;                                                       \
      case 112: SMALL_GO(KERNEL, 112, __VA_ARGS__); break;                                                       \
      default: SMALL_GO(KERNEL, 128, __VA_ARGS__); break;                                                        \
    }                                                                                                            \
  } while (0)

template <int HMAX>
static int launch_small_fwd_t(int B, int T, int H, const float* GI, int ldgi, const float* Whh, const float* bhh, float* Y,
                              float* gates, float* hprev, int hq, const float* labels, float* stat_part, hipStream_t st) {
  const size_t wbytes = (size_t)3 * H * H * sizeof(float);
  const int stage_w = wbytes <= 140 * 1024;
  const size_t smem = stage_w ? wbytes : 0;
  static std::atomic<unsigned long long> done{0};
  constexpr int KSP = small_ksp(HMAX);
  if (ensure_dyn_smem((const void*)gru_small_fwd_kernel<HMAX, 1, KSP>, 140 * 1024, done) != WGNN_OK) return WGNN_ERR_HIP;
  hipLaunchKernelGGL((gru_small_fwd_kernel<HMAX, 1, KSP>), dim3(B), dim3(KSP * cdiv_i(3 * H, 64) * 64), smem, st, B, T, H, GI,
                     ldgi, Whh, bhh, Y, gates, stage_w, hprev, hq, labels, stat_part);
  return WGNN_OK;
}

int launch_gru_small_fwd(int B, int T, int H, const float* GI, int ldgi, const float* Whh, const float* bhh, float* Y,
                         float* gates, float* hprev, int hq, const float* labels, float* stat_part, hipStream_t st) {
  if (!gru_small_supported(B, H)) return WGNN_ERR_UNSUPPORTED;
  if (hprev && (hq < H + 1 || hq % 4 != 0)) return WGNN_ERR_SHAPE;
  const double bt = (double)B * T;
  int rc = WGNN_OK;
  PROF_LAUNCH("gru_small_fwd_kernel", bt * 2.0 * 3 * H * H, bt * 4.0 * (3 * H + H + (gates ? 4 * H : 0) + (hprev ? hq : 0)), st,
              rc = small_hmax(H) == 32   ? launch_small_fwd_t<32>(B, T, H, GI, ldgi, Whh, bhh, Y, gates, hprev, hq, labels, stat_part, st)
                   : small_hmax(H) == 64  ? launch_small_fwd_t<64>(B, T, H, GI, ldgi, Whh, bhh, Y, gates, hprev, hq, labels, stat_part, st)
                   : small_hmax(H) == 96  ? launch_small_fwd_t<96>(B, T, H, GI, ldgi, Whh, bhh, Y, gates, hprev, hq, labels, stat_part, st)
                   : small_hmax(H) == 108 ? launch_small_fwd_t<108>(B, T, H, GI, ldgi, Whh, bhh, Y, gates, hprev, hq, labels, stat_part, st)
                   : small_hmax(H) == 112 ? launch_small_fwd_t<112>(B, T, H, GI, ldgi, Whh, bhh, Y, gates, hprev, hq, labels, stat_part, st)
                                          : launch_small_fwd_t<128>(B, T, H, GI, ldgi, Whh, bhh, Y, gates, hprev, hq, labels, stat_part, st));
  if (rc != WGNN_OK) return rc;
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int gru_small_blocks(int B) { return B; }        // workgroups of the forward = MSE partial pairs it writes with labels

int launch_gru_small_bwd(int B, int T, int H, const float* Whh, const float* Y, const float* dY, const float* labels,
                         const float* gates, float* dGI, float* dGH, int ldd, const float* stat_part, int64_t n_loss,
                         float grad_scale, float* loss, unsigned* status, hipStream_t st) {
  if (!gru_small_supported(B, H)) return WGNN_ERR_UNSUPPORTED;
  if ((dY == nullptr) == (labels == nullptr)) return WGNN_ERR_SHAPE;
  if (stat_part && (!labels || !loss)) return WGNN_ERR_NULL;
  const double bt = (double)B * T;
  const float inv_n = 1.0f / (float)n_loss, coef = 2.0f * grad_scale / (float)n_loss;
  PROF_LAUNCH("gru_small_bwd_kernel", bt * 2.0 * 3 * H * H, bt * 4.0 * (4 * H + 2 * H + 6 * H), st,
              SMALL_DISPATCH(gru_small_bwd_kernel, B, T, H, Whh, Y, dY, gates, dGI, dGH, ldd, labels, coef, stat_part,
                             gru_small_blocks(B), inv_n, loss, status));
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
