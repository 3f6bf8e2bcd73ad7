// GRU recurrence over the window, forward and backward (BPTT), exact-fp32 MFMA, W_hh resident in REGISTERS.
//
// Reference: nn.GRU(gru_input, gru_hidden_dim, batch_first=True) called with h0 = 0 at
// src/step6_gcn_gru_combined_model.py:11,23 (torch gate order r,z,n):
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r*gh_n), h = (1-z) n + z h_prev
// with gi = W_ih g + b_ih (precomputed for every timestep by the input-projection GEMM) and
// gh = W_hh h_prev + b_hh computed here; and its BPTT (src/main.py:79).
//
// One workgroup owns 16 windows (one MFMA M tile) for all T steps, 8 waves; wave w owns hidden units [16w, 16w + 16) of
// all three gates, so the gate math is lane-local in the C layout of v_mfma_f32_16x16x4_f32 (bitwise an fp32 fmaf chain).
// Each wave keeps its slice of W_hh (forward: [k = h index][n = gate unit]; backward: W_hh^T, [k = gate row][n = hidden
// unit]) as MFMA B operands in VGPRs for the whole launch -- 3 x 26 / 78 registers at H = 102 -- so LDS carries only the
// 16 x H state (h_t, or dgh_t) between waves, double-buffered by step parity (one barrier per step).  The MFMA's k slots
// are free to permute as long as A and B agree: lane (m, kq) takes k = kq KS + ks in step ks, i.e. a lane's A operands
// of all steps are KS CONTIGUOUS floats of its state row and come in as a few wide LDS reads instead of one per MFMA.
// (Round 2's kernels kept W_hh in LDS: two 4-byte LDS reads per MFMA and two barriers per step, 125 / 176 us.)
//
// What else the kernels do, so that no launch-sized pass is left around them in the exact-fp32 step:
//   forward  -- with labels (wgnn_fwd_loss): per-workgroup sums of (h - label)^2 into the stash (the loss needs no pass over
//               Y and the labels); the B operand of the dW_hh GEMM, [Hprev | 1 | 0..] with 16-byte aligned rows, written
//               directly (was hprev_pad_kernel); last_only (wgnn_fwd_last): only h_{T-1} * mul + add leaves the chip;
//   backward -- with labels: dY = 2 (Y - labels) grad_scale / n formed from the h_prev it loads anyway (no dY tensor, no
//               mse_kernel), the loss finalised by workgroup 0; of dGH only the n third is stored (dGHn): its r and z thirds
//               equal dGI's and the dW_hh GEMM takes them from there (two-source A operand, gemm32.hip).
#include "common.h"

namespace {

constexpr int MB = 16;        // windows per workgroup (one MFMA M tile)
constexpr int NTHREADS = 512; // 8 waves -> up to 128 hidden units
typedef float f32x2 __attribute__((ext_vector_type(2)));

// N contiguous floats (N even) from LDS into registers with the widest reads the alignment allows
template <int N>
__device__ __forceinline__ void lds_row(const float* p, float (&a)[N]) {
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const f32x4 v = *(const f32x4*)(p + 4 * i);
      a[4 * i] = v[0]; a[4 * i + 1] = v[1]; a[4 * i + 2] = v[2]; a[4 * i + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const f32x2 v = *(const f32x2*)(p + 2 * i);
      a[2 * i] = v[0]; a[2 * i + 1] = v[1];
    }
  }
}

// gate stash: grux.hip's layout, [workgroup][t][wave][r | z | n | gh_n][lane] x float4 (the 4 window rows a lane owns):
// one 16-byte access per lane and component, 1 KB per wave-instruction
__host__ __device__ inline size_t gate_floats(int B, int T, int H) {
  return (size_t)((B + MB - 1) / MB) * T * ((H + 15) / 16) * 4 * 64 * 4;
}

template <int KS>   // k steps of 4 over the hidden index, 4 KS >= H, KS even
__global__ void __launch_bounds__(NTHREADS) gru_fwd_kernel(int B, int T, int H, const float* __restrict__ GI, int ldgi,
                                                           const float* __restrict__ Whh,
                                                           const float* __restrict__ bhh, float* __restrict__ Y,
                                                           float* __restrict__ gates, const float* __restrict__ Lab,
                                                           float* __restrict__ stat_part, float* __restrict__ hprev,
                                                           int hq, int last_only, float y_mul, float y_add) {
  constexpr int KP = 4 * KS, HS = KP + 4;          // row stride: 16-byte aligned rows
  __shared__ __attribute__((aligned(16))) float hbuf[2 * MB * HS];
  for (int i = threadIdx.x; i < 2 * MB * HS; i += NTHREADS) hbuf[i] = 0.f;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int j = 16 * wave + lm;
  const bool active = 16 * wave < H;               // wave-uniform
  const bool jv = j < H;
  const int jc = jv ? j : H - 1;
  const int b0 = blockIdx.x * MB;
  // tag behind the MSE partial pairs: set by wgnn_fwd_loss, cleared by a plain forward on the same stash
  if (stat_part && blockIdx.x == 0 && threadIdx.x == 0) stat_part[2 * gridDim.x] = Lab ? WGNN_STATS_TAG : 0.f;
  if (hprev) {
    // [Hprev | 1 | 0..] rows: the constant tail of every row (b, t) and the whole row (b, 0) (h_{-1} = 0); the body of
    // row (b, t + 1) is written with h_t below
    const int tail = hq - H;
    for (int q = threadIdx.x; q < MB * T * tail; q += NTHREADS) {
      const int m = q / (T * tail), rem = q % (T * tail), t = rem / tail, c = H + rem % tail;
      if (b0 + m < B) hprev[((size_t)(b0 + m) * T + t) * hq + c] = c == H ? 1.f : 0.f;
    }
    for (int q = threadIdx.x; q < MB * H; q += NTHREADS) {
      const int m = q / H, c = q % H;
      if (b0 + m < B) hprev[((size_t)(b0 + m) * T) * hq + c] = 0.f;
    }
  }

  float WB[3][KS];                                 // B operand of step ks: W_hh[gate H + j][k = lk KS + ks]
#pragma unroll
  for (int gate = 0; gate < 3; ++gate)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = lk * KS + ks;
      WB[gate][ks] = (jv && k < H) ? Whh[(size_t)(gate * H + j) * H + k] : 0.f;
    }
  const float bh_r = bhh[jc], bh_z = bhh[H + jc], bh_n = bhh[2 * H + jc];

  const float* GIw = GI + (size_t)b0 * T * ldgi;
  float* Yw = Y + (size_t)b0 * T * H;
  const float* Labw = Lab ? Lab + (size_t)b0 * T * H : nullptr;
  float* Hpw = hprev ? hprev + (size_t)b0 * T * hq : nullptr;
  const int NW = (H + 15) / 16;
  // three components per record, r | z | gh_n: the BPTT kernel recomputes n = tanh(gi_n + r gh_n) from GI's n third, which
  // stays in the stash (round 4, as in grux.hip: the recurrences are bound by their bytes)
  f32x4* gatesw = gates ? (f32x4*)gates + ((size_t)blockIdx.x * T * NW + wave) * 3 * 64 + lane : nullptr;
  int rowt[4];            // (local window row) * T, clamped to the last valid window
  bool rowok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * lk + r;
    rowok[r] = jv && b0 + m < B;
    rowt[r] = (b0 + m < B ? m : B - 1 - b0) * T;
  }
  float gi[3][4], gin[3][4], lab[4] = {0.f, 0.f, 0.f, 0.f}, labn[4] = {0.f, 0.f, 0.f, 0.f};
  auto load_gi = [&](int t, float (&dst)[3][4], float (&ldst)[4]) {
    const int tc = t < T ? t : T - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (rowt[r] + tc) * ldgi + jc;
      dst[0][r] = GIw[o];
      dst[1][r] = GIw[o + H];
      dst[2][r] = GIw[o + 2 * H];
      if (Lab) ldst[r] = Labw[(rowt[r] + tc) * H + jc];
    }
  };
  load_gi(0, gi, lab);
  float hold[4] = {0.f, 0.f, 0.f, 0.f};
  float ssum = 0.f, smax = 0.f;
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const float* hcur = hbuf + (t & 1) * MB * HS;          // h_{t-1}
    float* hnext = hbuf + ((t + 1) & 1) * MB * HS;         // h_t goes here
    load_gi(t + 1, gin, labn);                             // prefetch under this step's MFMAs
    float hnew[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      f32x4 ar, az, an;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ar[r] = gi[0][r] + bh_r;
        az[r] = gi[1][r] + bh_z;
        an[r] = bh_n;
      }
      float a[KS];
      lds_row<KS>(hcur + lm * HS + lk * KS, a);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        ar = mfma16(a[ks], WB[0][ks], ar);
        az = mfma16(a[ks], WB[1][ks], az);
        an = mfma16(a[ks], WB[2][ks], an);
      }
      f32x4 rg4, zg4, ng4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // hardware exp2 / rcp forms (|error| < 3e-7, as in grux.hip and gru_small.hip)
        const float rg = sigmoid_fast(ar[r]);
        const float zg = sigmoid_fast(az[r]);
        const float ng = tanh_fast(__builtin_fmaf(rg, an[r], gi[2][r]));   // the BPTT kernel recomputes exactly this
        rg4[r] = rg; zg4[r] = zg; ng4[r] = ng;
        hnew[r] = (1.f - zg) * ng + zg * hold[r];
        if (rowok[r]) {
          if (!last_only) Yw[(rowt[r] + t) * H + j] = hnew[r];
          else if (t == T - 1) Y[(size_t)(b0 + 4 * lk + r) * H + j] = hnew[r] * y_mul + y_add;
          if (Hpw && t + 1 < T) Hpw[(rowt[r] + t + 1) * hq + j] = hnew[r];
          if (Lab) {
            const float dl = hnew[r] - lab[r];
            ssum = fmaf(dl, dl, ssum);
            smax = fmaxf(smax, fabsf(dl));
          }
        }
        hold[r] = hnew[r];
      }
      if (gates) {
        f32x4* rec = gatesw + (size_t)t * NW * 3 * 64;
        rec[0] = rg4;
        rec[64] = zg4;
        rec[128] = an;
      }
      if (jv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hnext[(4 * lk + r) * HS + j] = hnew[r];
      }
    }
    __syncthreads();                               // h_t complete; everyone is done reading h_{t-1}
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) gi[q][r] = gin[q][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) lab[r] = labn[r];
  }
  if (Lab) {   // block partials in a fixed order: lanes (xor tree), then the 8 waves
    __shared__ float red[2][NTHREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ssum += __shfl_xor(ssum, o, 64);
      smax = fmaxf(smax, __shfl_xor(smax, o, 64));
    }
    if (lane == 0) {
      red[0][wave] = ssum;
      red[1][wave] = smax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f, m = 0.f;
#pragma unroll
      for (int w = 0; w < NTHREADS / 64; ++w) {
        s += red[0][w];
        m = fmaxf(m, red[1][w]);
      }
      stat_part[blockIdx.x] = s;
      stat_part[gridDim.x + blockIdx.x] = m;
    }
  }
}

// BPTT.  Per step (t descending), with dh = dY_t + dh_next:
//   dn = dh (1-z), dz = dh (hprev - n), dnt = dn (1-n^2), dr = dnt gh_n,
//   dar = dr r (1-r), daz = dz z (1-z);  dgi = [dar, daz, dnt], dgh = [dar, daz, dnt r]
//   dh_next = dh z + dgh W_hh
// Outputs for the GEMMs that follow: dGI rows [B*T][ldd] (3H layout, zero K padding) and dGHn = dnt r rows [B*T][hn].
template <int KS3>   // k steps of 4 over the gate-row index, 4 KS3 >= 3H, KS3 even
__global__ void __launch_bounds__(NTHREADS) gru_bwd_kernel(int B, int T, int H, const float* __restrict__ Whh,
                                                           const float* __restrict__ Y, const float* __restrict__ dY,
                                                           const float* __restrict__ Lab,
                                                           const float* __restrict__ gates,
                                                           const float* __restrict__ GIn, int ldgi,
                                                           float* __restrict__ dGI,
                                                           int ldd, float* __restrict__ dGN, int hn,
                                                           float* __restrict__ dGH /*nullable: full rows [B*T][ldd]*/,
                                                           const float* __restrict__ stat_part, int nstat, float inv_n,
                                                           float coef_lab, float* __restrict__ loss_out,
                                                           unsigned* status) {
  constexpr int KP = 4 * KS3, DS = KP + 4;
  __shared__ __attribute__((aligned(16))) float dbuf[2 * MB * DS];   // dgh rows [dar | daz | dnr | 0..], by step parity
  for (int i = threadIdx.x; i < 2 * MB * DS; i += NTHREADS) dbuf[i] = 0.f;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int j = 16 * wave + lm;
  const bool active = 16 * wave < H;
  const bool jv = j < H;
  const int jc = jv ? j : H - 1;
  const int b0 = blockIdx.x * MB;
  const int G3 = 3 * H;
  if (stat_part && blockIdx.x == 0) {   // loss = (sum of the forward's per-workgroup partial sums) / n, fixed order
    __shared__ float sred[NTHREADS / 64];
    float a = 0.f;
    for (int i = threadIdx.x; i < nstat; i += NTHREADS) a += stat_part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) sred[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      a = 0.f;
#pragma unroll
      for (int w = 0; w < NTHREADS / 64; ++w) a += sred[w];
      const bool tagged = stat_part[2 * nstat] == WGNN_STATS_TAG;
      loss_out[0] = tagged ? a * inv_n : __builtin_nanf("");     // no statistics of these labels in the stash: loud
      if (!tagged && status) atomicOr(status, WGNN_STATUS_NO_LOSS_STATS);
    }
  }
  {  // zero the K-padding columns [3H, ldd) of this workgroup's dGI rows and [H, hn) of its dGHn rows
    const int nrows = min(MB, B - b0) * T;
    const int npad = ldd - G3, npn = hn - H;
    for (int i = threadIdx.x; i < nrows * npad; i += NTHREADS)
      dGI[((size_t)b0 * T + i / npad) * ldd + G3 + i % npad] = 0.f;
    if (dGN)
      for (int i = threadIdx.x; i < nrows * npn; i += NTHREADS)
        dGN[((size_t)b0 * T + i / npn) * hn + H + i % npn] = 0.f;
    if (dGH)
      for (int i = threadIdx.x; i < nrows * npad; i += NTHREADS)
        dGH[((size_t)b0 * T + i / npad) * ldd + G3 + i % npad] = 0.f;
  }

  float WT[KS3];                                   // B operand of step ks: W_hh[k = lk KS3 + ks][j]
#pragma unroll
  for (int ks = 0; ks < KS3; ++ks) {
    const int k = lk * KS3 + ks;
    WT[ks] = (jv && k < G3) ? Whh[(size_t)k * H + j] : 0.f;
  }
  const int NW = (H + 15) / 16;
  const f32x4* gatesw = (const f32x4*)gates + ((size_t)blockIdx.x * T * NW + (active ? wave : 0)) * 3 * 64 + lane;
  const float* GIw = GIn + (size_t)b0 * T * ldgi + 2 * H;          // n third of this workgroup's GI rows (stash)
  const float* Yw = Y + (size_t)b0 * T * H;
  const float* dYw = dY ? dY + (size_t)b0 * T * H : nullptr;
  const float* Labw = Lab ? Lab + (size_t)b0 * T * H : nullptr;
  float* dGIw = dGI + (size_t)b0 * T * ldd;
  float* dGNw = dGN ? dGN + (size_t)b0 * T * hn : nullptr;
  float* dGHw = dGH ? dGH + (size_t)b0 * T * ldd : nullptr;
  int rowt[4];
  bool rowok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * lk + r;
    rowok[r] = jv && b0 + m < B;
    rowt[r] = (b0 + m < B ? m : B - 1 - b0) * T;
  }
  struct StepIn { float dy[4], r[4], z[4], n[4], ghn[4], hp[4]; };
  auto load_step = [&](int t, StepIn& s) {
    const int tc = t > 0 ? t : 0;
    const f32x4* rec = gatesw + (size_t)tc * NW * 3 * 64;
    const f32x4 r4 = rec[0], z4 = rec[64], g4 = rec[128];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bt = rowt[r] + tc;
      s.dy[r] = Lab ? Labw[bt * H + jc] : dYw[bt * H + jc];       // the label, or dY itself
      s.r[r] = r4[r];
      s.z[r] = z4[r];
      s.n[r] = GIw[bt * ldgi + jc];                                // gi_n: n itself is formed in the step
      s.ghn[r] = g4[r];
      const float hp = Yw[(bt - (tc > 0 ? 1 : 0)) * H + jc];
      s.hp[r] = tc > 0 ? hp : 0.f;
    }
  };
  StepIn cur, nxt;
  load_step(T - 1, cur);
  float ycur[4] = {0.f, 0.f, 0.f, 0.f};              // Y[b, t]: the h_prev of step t + 1
  if (Lab) {
#pragma unroll
    for (int r = 0; r < 4; ++r) ycur[r] = Yw[(rowt[r] + T - 1) * H + jc];
  }
  f32x4 dhn = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    float* ds = dbuf + (t & 1) * MB * DS;
    load_step(t - 1, nxt);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * lk + r;
        const float dyv = Lab ? (ycur[r] - cur.dy[r]) * coef_lab : cur.dy[r];
        const float dh = rowok[r] ? dyv + dhn[r] : 0.f;
        const float rg = cur.r[r], zg = cur.z[r];
        const float ng = tanh_fast(__builtin_fmaf(rg, cur.ghn[r], cur.n[r]));   // the forward's n, bit for bit
        const float dn = dh * (1.f - zg);
        const float dz = dh * (cur.hp[r] - ng);
        const float dnt = dn * (1.f - ng * ng);
        const float dr = dnt * cur.ghn[r];
        const float dar = dr * rg * (1.f - rg);
        const float daz = dz * zg * (1.f - zg);
        const float dnr = dnt * rg;
        acc[r] = dh * zg;
        if (jv) {
          ds[m * DS + j] = dar;
          ds[m * DS + H + j] = daz;
          ds[m * DS + 2 * H + j] = dnr;
          if (rowok[r]) {
            float* gi = dGIw + (size_t)(rowt[r] + t) * ldd;
            gi[j] = dar;
            gi[H + j] = daz;
            gi[2 * H + j] = dnt;
            if (dGNw) dGNw[(size_t)(rowt[r] + t) * hn + j] = dnr;
            if (dGHw) {                          // small B*T: the general GEMM wants dGH whole
              float* gh = dGHw + (size_t)(rowt[r] + t) * ldd;
              gh[j] = dar;
              gh[H + j] = daz;
              gh[2 * H + j] = dnr;
            }
          }
        }
      }
    }
    __syncthreads();
    if (active && t > 0) {
      // dgh W_hh over the 3H gate rows: four independent accumulator chains (one KS3-long dependent chain of 16x16x4
      // MFMAs is pure latency)
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
      const float* row = ds + lm * DS + lk * KS3;
      constexpr int CH = KS3 % 4 == 0 ? 4 : (KS3 % 6 == 0 ? 6 : 2);      // k steps per LDS read batch (divides KS3)
      static_assert(KS3 % CH == 0, "batching of the state-row reads");
#pragma unroll
      for (int c0 = 0; c0 < KS3; c0 += CH) {
        float a[CH];
        lds_row<CH>(row + c0, a);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int ks = c0 + u;
          if ((ks & 3) == 0) acc = mfma16(a[u], WT[ks], acc);
          else if ((ks & 3) == 1) a1 = mfma16(a[u], WT[ks], a1);
          else if ((ks & 3) == 2) a2 = mfma16(a[u], WT[ks], a2);
          else a3 = mfma16(a[u], WT[ks], a3);
        }
      }
      acc = (acc + a1) + (a2 + a3);
    }
    dhn = acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) ycur[r] = cur.hp[r];   // Y[b, t-1]
    cur = nxt;
  }
}

// smallest supported step count >= need (instantiations: even, dense around the 34-station shapes)
int pick_ks(int need, const int* list, int n) {
  for (int i = 0; i < n; ++i)
    if (list[i] >= need) return list[i];
  return -1;
}
const int FWD_KS[] = {4, 8, 12, 16, 20, 24, 26, 28, 32};
const int BWD_KS3[] = {12, 24, 36, 48, 60, 72, 78, 84, 96};

}  // namespace

bool gru_shape_supported(int H) { return H >= 1 && H <= 16 * (NTHREADS / 64); }
size_t gru_gates_floats(int B, int T, int H) { return gate_floats(B, T, H); }
int gru_blocks(int B) { return cdiv_i(B, MB); }
int gru_hn(int H) { return 4 * cdiv_i(H, 4); }          // row width of the dGHn rows (16-byte aligned rows)
int gru_msplit(int H) { return 4 * cdiv_i(2 * H, 4); }   // first GEMM row of the dGHn block in the dW_hh product

int launch_gru_fwd(int B, int T, int H, const float* GI, int ldgi, const float* Whh, const float* bhh, float* Y,
                   float* gates, const float* labels, float* stat_part, float* hprev, int hq, int last_only, float y_mul,
                   float y_add, hipStream_t st) {
  if (!gru_shape_supported(H)) return WGNN_ERR_UNSUPPORTED;
  if (last_only && (gates || labels || hprev)) return WGNN_ERR_UNSUPPORTED;
  if (hprev && (hq < H + 1 || hq % 4 != 0)) return WGNN_ERR_SHAPE;
  const int ks = pick_ks(cdiv_i(H, 4), FWD_KS, (int)(sizeof(FWD_KS) / sizeof(int)));
  const double bt = (double)B * T;
  const double fl = bt * 2.0 * 3 * H * H;
  const double by = bt * 4.0 * (3 * H + (last_only ? 0 : H) + (labels ? H : 0) + (hprev ? hq : 0)) +
                    (gates ? 3.0 * gate_floats(B, T, H) : 0.0);     // three of the buffer's four components are used
#define FCASE(K)                                                                                                  \
  case K:                                                                                                         \
    PROF_LAUNCH("gru_fwd_kernel<" #K ">", fl, by, st,                                                             \
                hipLaunchKernelGGL(gru_fwd_kernel<K>, dim3(cdiv_i(B, MB)), dim3(NTHREADS), 0, st, B, T, H, GI, ldgi, Whh, \
                                   bhh, Y, gates, labels, stat_part, hprev, hq, last_only, y_mul, y_add));        \
    break
  switch (ks) {
    FCASE(4); FCASE(8); FCASE(12); FCASE(16); FCASE(20); FCASE(24); FCASE(26); FCASE(28); FCASE(32);
    default: return WGNN_ERR_UNSUPPORTED;
  }
#undef FCASE
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

// exactly one of dY / labels is non-null.  labels: dY = (Y - labels) * 2 grad_scale / n_loss is formed in the kernel;
// with stat_part (the forward's partial sums + tag) workgroup 0 also writes loss[0] = mean((Y - labels)^2).
int launch_gru_bwd(int B, int T, int H, const float* Whh, const float* Y, const float* dY, const float* labels,
                   const float* gates, const float* GI /*the forward's GI rows [B*T][ldgi] (stash)*/, int ldgi, float* dGI, int ldd, float* dGHn, float* dGH, const float* stat_part,
                   int64_t n_loss, float grad_scale, float* loss, unsigned* status, hipStream_t st) {
  if (!gru_shape_supported(H)) return WGNN_ERR_UNSUPPORTED;
  if ((dY == nullptr) == (labels == nullptr) || ldd < 3 * H || (dGHn == nullptr) == (dGH == nullptr)) return WGNN_ERR_SHAPE;
  if (stat_part && (!labels || !loss)) return WGNN_ERR_NULL;
  if (!GI || ldgi < 3 * H) return WGNN_ERR_NULL;
  const int ks3 = pick_ks(cdiv_i(3 * H, 4), BWD_KS3, (int)(sizeof(BWD_KS3) / sizeof(int)));
  const int hn = gru_hn(H);
  const float inv_n = 1.0f / (float)n_loss, coef = 2.0f * grad_scale / (float)n_loss;
  const double bt = (double)B * T;
  const double fl = bt * 2.0 * 3 * H * H, by = bt * 4.0 * (H + H + 3 * H + (dGH ? 3 * H : H)) + 3.0 * gate_floats(B, T, H) +
                                          bt * 4.0 * H;   // + GI's n third
#define BCASE(K)                                                                                                  \
  case K:                                                                                                         \
    PROF_LAUNCH("gru_bwd_kernel<" #K ">", fl, by, st,                                                             \
                hipLaunchKernelGGL(gru_bwd_kernel<K>, dim3(cdiv_i(B, MB)), dim3(NTHREADS), 0, st, B, T, H, Whh, Y, dY,  \
                                   labels, gates, GI, ldgi, dGI, ldd, dGHn, hn, dGH, stat_part, gru_blocks(B), inv_n, coef,  \
                                   loss, status));                                                                      \
    break
  switch (ks3) {
    BCASE(12); BCASE(24); BCASE(36); BCASE(48); BCASE(60); BCASE(72); BCASE(78); BCASE(84); BCASE(96);
    default: return WGNN_ERR_UNSUPPORTED;
  }
#undef BCASE
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
