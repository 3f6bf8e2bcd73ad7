// GRU recurrence, split-fp16 ("f16x3") MFMA, weights resident in REGISTERS.
//
// Reference: nn.GRU (1 layer, batch_first, h0 = 0, gates r,z,n) at
// src/step6_gcn_gru_combined_model.py:11,23 and its BPTT (src/main.py:79).
//
// One workgroup = 16 windows for all T steps, 8 waves; wave w owns hidden units [16w, 16w+16) for
// all three gates, so the gate math is lane-local in the 16x16 C layout.  Each wave keeps its slice
// of W_hh (forward: B operand [k = h index][n = gate unit]; backward: W_hh^T slice
// [k = gate row][n = hidden unit]) as fp16 hi/lo MFMA fragments in VGPRs for the whole launch
// (96 / 80 registers at H = 102); LDS only carries the 16 x H state (h, or dgh in the backward)
// between waves, stored as fp16 hi/lo rows padded to a conflict-free stride.
// Backward values (dY ~ 1e-9) are multiplied by the power of two scales[0] on load and written
// SCALED to dGI/dGH; the weight-gradient GEMMs un-scale in their epilogues.
#include "common.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int MB = 16;
constexpr int NTHREADS = 512;
// 16-byte components of a gate-stash record: r | z | n | gh_n, plus h_t itself when Y on the wire is 16-bit (the BPTT
// kernel then takes h_{t-1} from the stash: the rounded Y is not the forward's hidden state)
__host__ __device__ constexpr int grec(int io) { return io ? 5 : 4; }

struct Frag { h8 hi, lo; };

__device__ __forceinline__ f32x4 mfma_x(h8 a, h8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// X3: split-fp16 three-pass product; !X3: plain fp16 operands, one pass ("f16" math mode)
template <bool X3>
__device__ __forceinline__ f32x4 mfma3(const Frag& a, const Frag& b, f32x4 c) {
  if (X3) {
    c = mfma_x(a.lo, b.hi, c);
    c = mfma_x(a.hi, b.lo, c);
  }
  c = mfma_x(a.hi, b.hi, c);
  return c;
}
// hi = fp16(x) (v_cvt_pk_f16_f32, 2 values per instruction), lo = fp16(x - hi) with the difference
// formed by v_fma_mix_f32 reading hi straight out of the packed register: 2 VALU per value instead
// of the 3 hipcc emits for the C expression (these kernels are VALU-issue-bound on exactly this).
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  float t0, t1;
  asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_cvt_pk_f16_f32 %1, %2, %3\n\t"
      "s_nop 1"   // VALU write -> MFMA operand read needs 2 wait states; hipcc pads nothing inside/after asm
      : "=&v"(hi), "=&v"(lo), "=&v"(t0), "=&v"(t1)
      : "v"(x0), "v"(x1));
}
__device__ __forceinline__ Frag split_vals(const float (&x)[8]) {   // weights: once per launch, both halves
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  u32x4v hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h, l;
    split2(x[2 * j], x[2 * j + 1], h, l);
    hi[j] = h;
    lo[j] = l;
  }
  Frag f;
  f.hi = __builtin_bit_cast(h8, hi);
  f.lo = __builtin_bit_cast(h8, lo);
  return f;
}
// io-typed element access (wgnn_io): 0 = fp32, 1 = fp16, 2 = bf16 (round to nearest even on store)
__device__ __forceinline__ float io_load(const void* p, int idx, int io) {
  if (io == 0) return ((const float*)p)[idx];
  const unsigned short u = ((const unsigned short*)p)[idx];
  if (io == 1) return (float)__builtin_bit_cast(_Float16, u);
  return __builtin_bit_cast(float, (unsigned)u << 16);
}
__device__ __forceinline__ void io_store(void* p, int idx, float v, int io) {
  if (io == 0) { ((float*)p)[idx] = v; return; }
  if (io == 1) { ((_Float16*)p)[idx] = (_Float16)v; return; }
  ((__bf16*)p)[idx] = (__bf16)v;      // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
}
template <bool X3>
__device__ __forceinline__ void put_split(_Float16* hi, _Float16* lo, int idx, float v) {
  const _Float16 h = (_Float16)v;
  hi[idx] = h;
  if (X3) lo[idx] = (_Float16)(v - (float)h);
}

// ------------------------------------------------------------------------------------------------
// IO: Y and the labels are 16-bit on the wire (fp16 / bf16 by `io`) and move through LDS as whole rows; the fp32
// instance keeps the direct per-lane accesses (the row path was 30 % slower there: measured 116 vs 84 us).
// GI: the input projection [B*T][ldgi] in the 3H layout -- fp32 for the split (X3) instances, ONE fp16 plane for the one-pass
// fp16 instances (WGNN_MATH_F16: pgemm_nt's OUT16 epilogue; half the bytes of the largest intermediate of that mode)
template <int KS, bool X3, bool IO>   // K steps of 32 over the hidden index (+ the ones column): KS = ceil((H+1)/32)
__global__ void __launch_bounds__(NTHREADS) grux_fwd_kernel(int B, int T, int H, const void* __restrict__ GI, int ldgi,
                                                            const float* __restrict__ Whh,
                                                            const float* __restrict__ bhh, void* __restrict__ Y,
                                                            float* __restrict__ gates, _Float16* __restrict__ yp_hi,
                                                            _Float16* __restrict__ yp_lo, unsigned* status,
                                                            const void* __restrict__ Lab,
                                                            float* __restrict__ stat_part, int io, int last_only,
                                                            float y_mul, float y_add) {
  // last_only (wgnn_fwd_last, inference): Y is [B][H] fp32 and receives only h_{T-1} * y_mul + y_add -- the evaluation
  // read-out of src/main.py:103,116 without writing (and re-reading) the other T-1 rows
  constexpr int HP = 32 * KS;                      // plane row width (halfs): h, then 1.0 at column H, then 0
  constexpr int HS = 32 * KS + 8;                  // row stride in halfs: 16-B aligned, conflict-free reads
  // h state is double-buffered: step t reads buffer t&1 and writes h_t into the other one, so a single
  // barrier per step suffices (the recurrence is latency-bound: every barrier is on the critical path)
  __shared__ __attribute__((aligned(16))) _Float16 hbuf[2 * 2 * MB * HS];
  for (int i = threadIdx.x; i < 2 * 2 * MB * HS; i += NTHREADS) hbuf[i] = (_Float16)0.f;
  __syncthreads();
  if (threadIdx.x < 2 * MB)   // ones column of both buffers' hi planes (W_hh fragments are 0 there)
    hbuf[(threadIdx.x / MB) * 2 * MB * HS + (threadIdx.x % MB) * HS + H] = (_Float16)1.f;

  // tag behind the MSE partial pairs: set by wgnn_fwd_loss, cleared by a plain wgnn_fwd on the same stash, checked by the
  // BPTT kernel when the caller claims (part bit 8) that the statistics of THESE labels are in the stash
  if (stat_part && blockIdx.x == 0 && threadIdx.x == 0) stat_part[2 * gridDim.x] = Lab ? WGNN_STATS_TAG : 0.f;
  if (yp_hi && blockIdx.x == 0 && threadIdx.x < HP) {   // row B*T: what [Hprev | 1] looks like at t = 0
    yp_hi[(size_t)B * T * HP + threadIdx.x] = (_Float16)(threadIdx.x == H ? 1.f : 0.f);
    yp_lo[(size_t)B * T * HP + threadIdx.x] = (_Float16)0.f;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int j = 16 * wave + c;
  const bool active = 16 * wave < H;               // wave-uniform
  const bool jv = j < H;
  const int jc = jv ? j : H - 1;
  const int b0 = blockIdx.x * MB;

  bool wbad = false;
  Frag WB[3][KS];                                  // B operand: W_hh[(gate*H + j)][k], k = 32ks + 8g + jj
#pragma unroll
  for (int gate = 0; gate < 3; ++gate)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float x[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int k = 32 * ks + 8 * g + jj;
        x[jj] = (jv && k < H) ? Whh[(size_t)(gate * H + j) * H + k] : 0.f;
        wbad |= out_of_fp16_range(x[jj]);
      }
      WB[gate][ks] = split_vals(x);
    }
  if (blockIdx.x == 0) report_status(status, wbad, WGNN_STATUS_WEIGHT_RANGE);
  const float bh_r = bhh[jc], bh_z = bhh[H + jc], bh_n = bhh[2 * H + jc];

  // Addressing: one workgroup-uniform 64-bit base per array (the workgroup's first window) plus 32-bit lane
  // offsets.  Row indices are clamped so every load is unconditional.
  typedef typename std::conditional<X3, float, _Float16>::type gi_t;
  const gi_t* GIw = (const gi_t*)GI + (size_t)b0 * T * ldgi;
  constexpr int esz = IO ? 2 : 4;                  // bytes per element of Y and the labels (wgnn_io)
  void* Yw = (char*)Y + (size_t)b0 * T * H * esz;
  // gate stash in this kernel's own register layout, [workgroup][t][wave][r | z | n | gh_n][lane] x float4 (the 4 window
  // rows a lane owns): one 16-byte store per lane and component (a full 1 KB per wave-instruction) instead of sixteen
  // 4-byte stores in 64-byte segments; the BPTT kernel (same thread <-> (rows, unit) map) reads it back the same way.
  // All four stay fp32: packing r and z as unorm16 (12 bytes per record: forward -4 us, BPTT -11 us) was tried and is
  // WRONG for trained weights -- saturated gates need r (1 - r) to fp32's own precision (8 % gradient error on the
  // wind_gnn_7.pth fixture).
  const int NW = (H + 15) / 16;
  // One-pass fp16 mode (X3 = false; tolerance class 5e-2): the record is fp16 -- [r | z] and [n | gh_n] as two 8-half
  // (16-byte) components -- half the stash bytes of the two recurrences, which are bound by exactly those bytes at the bench
  // size.  (Evaluated on the fp64 oracle in round 3: gradient errors 0.9-2.8e-4 of max, independent of B: outside the 1e-4
  // bar of the split modes, far inside this mode's.)  The h_t component of 16-bit I/O stays fp32.
  // Split modes (round 4): THREE fp32 components, r | z | gh_n -- n is not stored: the BPTT kernel recomputes it as
  // tanh(gi_n + r gh_n) from the n third of GI, which the forward keeps in the stash (the projection GEMM writes GI there
  // instead of into the workspace: no extra traffic); 44 MB less stash per step at B = 4096 for 42 MB of GI read back.
  constexpr int GREC = X3 ? (IO ? 4 : 3) : (IO ? 3 : 2);       // 16-byte components per lane and (t, wave)
  f32x4* gatesw = gates ? (f32x4*)gates + ((size_t)blockIdx.x * T * NW + wave) * GREC * 64 + lane : nullptr;
  int rowt[4];            // (local window row) * T, clamped to the last valid window
  bool rowok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * g + r;
    rowok[r] = jv && b0 + m < B;
    rowt[r] = (b0 + m < B ? m : B - 1 - b0) * T;
  }
  // ---- Y out and labels in go through LDS as whole rows: a row (b, t) of Y or of the labels is H contiguous elements,
  // moved as "units" of 2 elements (8 bytes fp32, 4 bytes fp16 / bf16) by consecutive threads, instead of each owner
  // lane storing / loading single elements in 64-byte (32-byte at 16 bits) segments.  ytile / ltile are double-buffered
  // by step parity like hbuf.  Unit q = threadIdx.x + 512 i of the workgroup's 16 rows: fixed for the whole launch.
  constexpr int HY = HP + 2;                       // fp32 row stride of the tiles (even: 8-byte aligned pairs)
  __shared__ __attribute__((aligned(16))) float ytile[IO ? 2 * MB * HY : 4];
  __shared__ __attribute__((aligned(16))) float ltile[IO ? 2 * MB * HY : 4];
  const int UPR = (H + 1) / 2;                     // units per row
  constexpr int NU = (MB * (HP / 2) + NTHREADS - 1) / NTHREADS;
  int u_lds[NU], u_glb[NU], u_n[NU];               // tile offset (m * HY + e), row-relative global element offset, elements (0, 1, 2)
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int q = threadIdx.x + NTHREADS * i;
    const int m = q / UPR, e = 2 * (q % UPR);
    const bool ok = q < MB * UPR && b0 + m < B;
    u_lds[i] = ok ? m * HY + e : 0;
    u_glb[i] = ok ? m * T * H + e : 0;
    u_n[i] = ok ? (e + 1 < H ? 2 : 1) : 0;
  }
  // wgnn_fwd_loss: the MSE statistics (sum of squares, max |Y - L|) are taken here, from the h this kernel
  // holds in registers, so no later pass re-reads Y and the labels for them
  const void* Labw = Lab ? (const char*)Lab + (size_t)b0 * T * H * esz : nullptr;
  f32x2 lraw[NU];                                  // the next step's label units, in flight (raw bits when 16-bit)
  auto load_lab = [&](int t) {
    if (!IO || !Lab) return;
    const int tc = t < T ? t : T - 1;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int o = u_glb[i] + tc * H;
      if (io == 0) {
        if (u_n[i] == 2) lraw[i] = *(const f32x2*)((const float*)Labw + o);
        else if (u_n[i] == 1) lraw[i][0] = ((const float*)Labw)[o];
      } else {
        if (u_n[i] == 2) lraw[i][0] = __builtin_bit_cast(float, *(const unsigned*)((const unsigned short*)Labw + o));
        else if (u_n[i] == 1) lraw[i][0] = __builtin_bit_cast(float, (unsigned)((const unsigned short*)Labw)[o]);
      }
    }
  };
  auto stage_lab = [&](int t) {                    // labels of step t -> ltile[t & 1] as fp32
    if (!IO || !Lab) return;
    float* lt = ltile + (t & 1) * MB * HY;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      if (u_n[i] == 0) continue;
      f32x2 v = lraw[i];
      if (io == 1) {
        v = __builtin_convertvector(__builtin_bit_cast(h2v, __builtin_bit_cast(unsigned, lraw[i][0])), f32x2);
      } else if (io == 2) {
        const unsigned u = __builtin_bit_cast(unsigned, lraw[i][0]);
        v[0] = __builtin_bit_cast(float, u << 16);
        v[1] = __builtin_bit_cast(float, u & 0xffff0000u);
      }
      lt[u_lds[i]] = v[0];
      if (u_n[i] == 2) lt[u_lds[i] + 1] = v[1];
    }
  };
  float ssum = 0.f, smax = 0.f;
  float lab[4] = {0.f, 0.f, 0.f, 0.f}, labn[4] = {0.f, 0.f, 0.f, 0.f};   // fp32 I/O: the owner lanes load their labels
  float gi[3][4], gin[3][4];
  auto load_gi = [&](int t, float (&dst)[3][4], float (&ldst)[4]) {
    const int tc = t < T ? t : T - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (rowt[r] + tc) * ldgi + jc;
      dst[0][r] = (float)GIw[o];
      dst[1][r] = (float)GIw[o + H];
      dst[2][r] = (float)GIw[o + 2 * H];
      if (!IO && Lab) ldst[r] = ((const float*)Labw)[(rowt[r] + tc) * H + jc];
    }
  };
  load_lab(0);
  stage_lab(0);
  load_gi(0, gi, lab);
  float hold[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const _Float16* hhi = hbuf + (t & 1) * 2 * MB * HS;          // h_{t-1}
    const _Float16* hlo = hhi + MB * HS;
    _Float16* nhi = hbuf + ((t + 1) & 1) * 2 * MB * HS;          // h_t goes here
    _Float16* nlo = nhi + MB * HS;
    load_gi(t + 1, gin, labn);                     // prefetch under this step's MFMAs
    load_lab(t + 1);
    float* yt = ytile + (t & 1) * MB * HY;
    const float* lt = ltile + (t & 1) * MB * HY;
    float hnew[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      f32x4 ar, az, an;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ar[r] = gi[0][r] + bh_r;
        az[r] = gi[1][r] + bh_z;
        an[r] = bh_n;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Frag a;
        a.hi = *(const h8*)(hhi + c * HS + 32 * ks + 8 * g);
        if (X3) a.lo = *(const h8*)(hlo + c * HS + 32 * ks + 8 * g);
        else a.lo = a.hi;
        ar = mfma3<X3>(a, WB[0][ks], ar);
        az = mfma3<X3>(a, WB[1][ks], az);
        an = mfma3<X3>(a, WB[2][ks], an);
      }
      f32x4 rg4, zg4, ng4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {                      // the four rows' gate chains in ONE basic block (they interleave);
        const float rg = sigmoid_fast(ar[r]);            // everything conditional comes in the loop below
        const float zg = sigmoid_fast(az[r]);
        const float ng = tanh_fast(__builtin_fmaf(rg, an[r], gi[2][r]));   // the BPTT kernel recomputes exactly this
        rg4[r] = rg; zg4[r] = zg; ng4[r] = ng;
        hnew[r] = (1.f - zg) * ng + zg * hold[r];
        hold[r] = hnew[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (IO) {
          if (jv) yt[(4 * g + r) * HY + j] = hnew[r];   // -> Y through the row copy-out below
        } else if (rowok[r]) {
          if (!last_only) ((float*)Yw)[(rowt[r] + t) * H + j] = hnew[r];
          else if (t == T - 1) ((float*)Y)[(size_t)(b0 + 4 * g + r) * H + j] = hnew[r] * y_mul + y_add;
        }
        if (rowok[r] && Lab) {                          // the statistics use the unrounded h
          const float dl = hnew[r] - (IO ? lt[(4 * g + r) * HY + j] : lab[r]);
          ssum = fmaf(dl, dl, ssum);
          smax = fmaxf(smax, fabsf(dl));
        }
      }
      if (gates) {
        f32x4* rec = gatesw + (size_t)t * NW * GREC * 64;
        if (X3) {
          rec[0] = rg4;
          rec[64] = zg4;
          rec[128] = an;
        } else {
          h8 rz, ng;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            rz[r] = (_Float16)rg4[r];
            rz[4 + r] = (_Float16)zg4[r];
            ng[r] = (_Float16)ng4[r];
            ng[4 + r] = (_Float16)an[r];
          }
          rec[0] = __builtin_bit_cast(f32x4, rz);
          rec[64] = __builtin_bit_cast(f32x4, ng);
        }
        if (IO) {
          const f32x4 h4 = {hnew[0], hnew[1], hnew[2], hnew[3]};
          rec[(GREC - 1) * 64] = h4;
        }
      }
    }
    if (active) {                                    // lanes past H aim at the row's pad halfs (never read as K, never copied out)
#pragma unroll
      for (int r = 0; r < 4; ++r) put_split<X3>(nhi, nlo, (4 * g + r) * HS + (jv ? j : HP + (c & 7)), hnew[r]);
    }
    stage_lab(t + 1);                              // other parity than the labels read above
    __syncthreads();                               // h_t complete; everyone is done reading h_{t-1}
#pragma unroll
    for (int i = 0; i < (IO ? NU : 0); ++i) {      // row (b, t) of Y, 2 elements per thread, rounded once to the I/O type
      if (u_n[i] == 0) continue;
      const float v0 = yt[u_lds[i]], v1 = yt[u_lds[i] + 1];
      const int o = u_glb[i] + t * H;
      if (io == 0) {
        if (u_n[i] == 2) { const f32x2 v = {v0, v1}; *(f32x2*)((float*)Yw + o) = v; }
        else ((float*)Yw)[o] = v0;
      } else {
        unsigned pk;
        if (io == 1) {
          const f32x2 v = {v0, v1};
          pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2v));
        } else {
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          const f32x2 v = {v0, v1};
          pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
        }
        if (u_n[i] == 2) *(unsigned*)((unsigned short*)Yw + o) = pk;
        else ((unsigned short*)Yw)[o] = (unsigned short)pk;
      }
    }
    if (yp_hi) {   // h_t as fp16 planes (the B operand of the dW_hh GEMM): 16-byte chunks straight from LDS
      for (int q = threadIdx.x; q < (X3 ? 2 : 1) * MB * (HP / 8); q += NTHREADS) {
        const int plane = q / (MB * (HP / 8)), rem = q % (MB * (HP / 8));
        const int m = rem / (HP / 8), ch = rem % (HP / 8);
        const int b = b0 + m;
        if (b < B) {
          const h8 v = *(const h8*)((plane ? nlo : nhi) + m * HS + 8 * ch);
          // non-temporal: the planes are next read by the dW_hh GEMM, four kernels later; kept out of the Infinity Cache they
          // leave it to the gate stash and Y, which the BPTT kernel reads back right after this one
          __builtin_nontemporal_store(v, (h8*)((plane ? yp_lo : yp_hi) + (size_t)b0 * T * HP + (m * T + t) * HP + 8 * ch));
        }
      }
    }
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
      float a = 0.f, m = 0.f;
#pragma unroll
      for (int w = 0; w < NTHREADS / 64; ++w) {
        a += red[0][w];
        m = fmaxf(m, red[1][w]);
      }
      stat_part[blockIdx.x] = a;
      stat_part[gridDim.x + blockIdx.x] = m;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// BPTT; per step (t descending), dh = s*dY_t + dh_next (everything in units scaled by s = scales[0]):
//   dn = dh (1-z), dz = dh (hprev - n), dnt = dn (1-n^2), dr = dnt gh_n,
//   dar = dr r (1-r), daz = dz z (1-z);  dgi = [dar, daz, dnt], dgh = [dar, daz, dnt r]
//   dh_next = dh z + dgh W_hh
// Outputs for the weight-gradient GEMMs: dGI planes [B*T][ldd] in the 3H layout, and of dGH ONLY its n third,
// dGHn = dnt*r, planes [B*T][HN] (dGH's r and z thirds equal dGI's: the dW_hh GEMM takes them from the dGI planes).
// In LDS the dgh row is laid out [dar | daz | pad to MS = 8*ceil(2H/8) | dnr] so that the dnr block is 16-byte aligned
// for the copy-out; the W_hh^T fragments' k index follows the same layout.
template <int KSB, bool X3, bool IO>   // K steps of 32 over the padded dgh row: KSB = ceil((MS + H) / 32); IO: as grux_fwd_kernel
__global__ void __launch_bounds__(NTHREADS) grux_bwd_kernel(int B, int T, int H, const float* __restrict__ Whh,
                                                            const void* __restrict__ Y, const float* __restrict__ dY,
                                                            const void* __restrict__ Lab, int io,
                                                            const float* __restrict__ gates,
                                                            const float* __restrict__ GIn, int ldgi,
                                                            const float* __restrict__ scales,
                                                            _Float16* __restrict__ dGI_hi, _Float16* __restrict__ dGI_lo,
                                                            int ldd, _Float16* __restrict__ dGN_hi,
                                                            _Float16* __restrict__ dGN_lo,
                                                            const float* __restrict__ stat_part, int nstat, float inv_n,
                                                            float coef_in, float* __restrict__ loss_out,
                                                            float* __restrict__ scales_out, unsigned* status,
                                                            int write_lo) {
  // write_lo == 0 (X3, large B*T): dGI / dGHn leave the chip as ONE fp16 plane (the lo halves stay in LDS, where the
  // recurrence's own product dgh W_hh uses them): the three GEMMs that consume them run two passes
  constexpr int DS = 32 * KSB + 24;   // row stride in halfs: odd in 16-byte units (conflict-free b128 rows); 8 zero pad halfs, then 8 dump halfs
  // stat_part != null (wgnn_bwd_mse_part(.. | 8) after wgnn_fwd_loss): the loss, the power-of-two range scale and the dY
  // coefficient are finalised HERE from the forward recurrence's nstat partial pairs (sum | max), by every workgroup for
  // itself (2 nstat floats out of L2; the max is order-independent, so all workgroups agree on the scale), instead of in a
  // launch of their own; workgroup 0 publishes loss and scales for the kernels that follow.
  float s_fin = 1.f, c_fin = coef_in;
  if (stat_part) {
    __shared__ float sred[2][NTHREADS / 64];
    float a = 0.f, m = 0.f;
    for (int i = threadIdx.x; i < nstat; i += NTHREADS) {
      a += stat_part[i];
      m = fmaxf(m, stat_part[nstat + i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      m = fmaxf(m, __shfl_xor(m, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      sred[0][threadIdx.x >> 6] = a;
      sred[1][threadIdx.x >> 6] = m;
    }
    __syncthreads();
    a = 0.f;
    m = 0.f;
#pragma unroll
    for (int w = 0; w < NTHREADS / 64; ++w) {
      a += sred[0][w];
      m = fmaxf(m, sred[1][w]);
    }
    const bool tagged = stat_part[2 * nstat] == WGNN_STATS_TAG;
    m *= fabsf(coef_in);
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);
      s_fin = ldexpf(1.f, 1 - e);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      loss_out[0] = tagged ? a * inv_n : __builtin_nanf("");     // no statistics of these labels in the stash: loud, not stale
      scales_out[0] = s_fin;
      scales_out[1] = 1.f / s_fin;
      scales_out[2] = coef_in;
      if (!tagged && status) atomicOr(status, WGNN_STATUS_NO_LOSS_STATS);
    }
  }
  // per step parity: dgh hi | dgh lo (padded layout, also the MFMA A operand) | dgi hi | dgi lo (3H layout); columns
  // never written (K padding, the tail of the dnr block) stay zero.  Double-buffered by step parity (kept with
  // the second barrier below: dropping that barrier was slower)
  __shared__ __attribute__((aligned(16))) _Float16 dbuf2[2 * 4 * MB * DS];
  for (int i = threadIdx.x; i < 2 * 4 * MB * DS; i += NTHREADS) dbuf2[i] = (_Float16)0.f;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int j = 16 * wave + c;
  const bool active = 16 * wave < H;
  const bool jv = j < H;
  const int jc = jv ? j : H - 1;
  const int b0 = blockIdx.x * MB;
  const int MS = 8 * ((2 * H + 7) / 8), HN = 8 * ((H + 7) / 8);
  const float s_in = stat_part ? s_fin : (scales ? scales[0] : 1.f);
  // fused MSE (wgnn_bwd_mse_part): dY is not materialised, dY[b,t] = (Y[b,t] - L[b,t]) * scales[2] is formed here;
  // Y[b,t] is the h_prev this kernel loaded for step t+1, so only L is read in dY's place.
  const float coef = stat_part ? c_fin : (Lab ? scales[2] : 1.f);

  Frag WT[KSB];                                    // B operand: W_hh[row(k)][j], k = 32ks + 8g + jj in the padded layout
#pragma unroll
  for (int ks = 0; ks < KSB; ++ks) {
    float x[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int k = 32 * ks + 8 * g + jj;
      const int row = k < 2 * H ? k : ((k >= MS && k - MS < H) ? 2 * H + (k - MS) : -1);
      x[jj] = (jv && row >= 0) ? Whh[(size_t)row * H + j] : 0.f;
    }
    WT[ks] = split_vals(x);
  }
  const int NW = (H + 15) / 16;                              // gate stash: grux_fwd_kernel's register layout
  const f32x4* gatesw = (const f32x4*)gates + ((size_t)blockIdx.x * T * NW + (active ? wave : 0)) * (X3 ? (IO ? 4 : 3) : (IO ? 3 : 2)) * 64 + lane;
  // 16-bit I/O: Y on the wire is rounded, so h_{t-1} (and the Y of the fused dY) is the 5th component of the stash
  // records; explicit dY is always fp32; labels are io-typed and come in through LDS as whole rows (see grux_fwd_kernel)
  constexpr int esz = IO ? 2 : 4;
  constexpr int GREC = X3 ? (IO ? 4 : 3) : (IO ? 3 : 2);       // as grux_fwd_kernel: r | z | gh_n (+ h), or fp16 records
  const float* GIw = X3 ? GIn + (size_t)b0 * T * ldgi + 2 * H : nullptr;   // n third of this workgroup's GI rows (stash)
  const void* Labw = Lab ? (const char*)Lab + (size_t)b0 * T * H * esz : nullptr;
  const float* dYw = dY ? dY + (size_t)b0 * T * H : nullptr;
  const float* Yw = (const float*)Y + (size_t)b0 * T * H;
  int rowt[4];
  bool rowok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * g + r;
    rowok[r] = jv && b0 + m < B;
    rowt[r] = (b0 + m < B ? m : B - 1 - b0) * T;
  }
  // ---- copy-out plan, fixed for the whole launch: per step the workgroup moves, as 16-byte chunks,
  // 2 planes x 16 rows x ldd/8 chunks of the dgi tile and 2 planes x 16 rows x HN/8 chunks of the dnr block.
  // Chunk q = threadIdx.x + 512 i: its LDS offset, destination and row stride never change (only t does).
  constexpr int NCH = (2 * MB * (4 * KSB) + 2 * MB * (4 * KSB) / 2 + NTHREADS - 1) / NTHREADS;   // upper bound
  int c_lds[NCH], c_step[NCH];
  _Float16* c_dst[NCH];
  {
    const int cpr = ldd / 8, cpn = HN / 8;
    const int n_i = 2 * MB * cpr, n_n = 2 * MB * cpn;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int q = threadIdx.x + NTHREADS * i;
      c_dst[i] = nullptr;
      c_lds[i] = 0;
      c_step[i] = 0;
      if (q < n_i) {
        const int plane = q / (MB * cpr), rem = q % (MB * cpr), m = rem / cpr, ch = rem % cpr;
        if (b0 + m < B && ((X3 && write_lo) || plane == 0)) {
          c_lds[i] = (2 + plane) * MB * DS + m * DS + 8 * ch;
          c_dst[i] = (plane ? dGI_lo : dGI_hi) + ((size_t)(b0 + m) * T) * ldd + 8 * ch;
          c_step[i] = ldd;
        }
      } else if (q - n_i < n_n) {
        q -= n_i;
        const int plane = q / (MB * cpn), rem = q % (MB * cpn), m = rem / cpn, ch = rem % cpn;
        if (b0 + m < B && ((X3 && write_lo) || plane == 0)) {
          c_lds[i] = plane * MB * DS + m * DS + MS + 8 * ch;
          c_dst[i] = (plane ? dGN_lo : dGN_hi) + ((size_t)(b0 + m) * T) * HN + 8 * ch;
          c_step[i] = HN;
        }
      }
    }
  }
  constexpr int HY = 32 * ((KSB * 32 / 3 + 31) / 32) + 34;     // >= H + 2 (H <= (32 KSB) / 3 + a few), even
  __shared__ __attribute__((aligned(16))) float ltile[IO ? 2 * MB * HY : 4];
  const int UPR = (H + 1) / 2;
  constexpr int NU = IO ? (MB * (HY / 2) + NTHREADS - 1) / NTHREADS : 1;
  int u_lds[NU], u_glb[NU], u_n[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int q = threadIdx.x + NTHREADS * i;
    const int m = q / UPR, e = 2 * (q % UPR);
    const bool ok = IO && Lab && q < MB * UPR && b0 + m < B;
    u_lds[i] = ok ? m * HY + e : 0;
    u_glb[i] = ok ? m * T * H + e : 0;
    u_n[i] = ok ? (e + 1 < H ? 2 : 1) : 0;
  }
  f32x2 lraw[NU];
  auto load_lab = [&](int t) {
    const int tc = t > 0 ? t : 0;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int o = u_glb[i] + tc * H;
      if (io == 0) {
        if (u_n[i] == 2) lraw[i] = *(const f32x2*)((const float*)Labw + o);
        else if (u_n[i] == 1) lraw[i][0] = ((const float*)Labw)[o];
      } else {
        if (u_n[i] == 2) lraw[i][0] = __builtin_bit_cast(float, *(const unsigned*)((const unsigned short*)Labw + o));
        else if (u_n[i] == 1) lraw[i][0] = __builtin_bit_cast(float, (unsigned)((const unsigned short*)Labw)[o]);
      }
    }
  };
  auto stage_lab = [&](int t) {                    // labels of step t -> ltile[t & 1] as fp32
    float* lt = ltile + (t & 1) * MB * HY;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      if (u_n[i] == 0) continue;
      f32x2 v = lraw[i];
      if (io == 1) {
        v = __builtin_convertvector(__builtin_bit_cast(h2v, __builtin_bit_cast(unsigned, lraw[i][0])), f32x2);
      } else if (io == 2) {
        const unsigned u = __builtin_bit_cast(unsigned, lraw[i][0]);
        v[0] = __builtin_bit_cast(float, u << 16);
        v[1] = __builtin_bit_cast(float, u & 0xffff0000u);
      }
      lt[u_lds[i]] = v[0];
      if (u_n[i] == 2) lt[u_lds[i] + 1] = v[1];
    }
  };
  struct StepIn { float dy[4], r[4], z[4], n[4], ghn[4], hp[4]; };
  auto load_step = [&](int t, StepIn& s) {
    const int tc = t > 0 ? t : 0;
    const f32x4* rec = gatesw + (size_t)tc * NW * GREC * 64;
    f32x4 r4, z4, n4, g4;
    if (X3) {
      r4 = rec[0]; z4 = rec[64]; g4 = rec[128];
#pragma unroll
      for (int r = 0; r < 4; ++r) n4[r] = GIw[(rowt[r] + tc) * ldgi + jc];      // gi_n: n itself is formed in the step
    } else {
      const h8 rz = __builtin_bit_cast(h8, rec[0]), ng = __builtin_bit_cast(h8, rec[64]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        r4[r] = (float)rz[r];
        z4[r] = (float)rz[4 + r];
        n4[r] = (float)ng[r];
        g4[r] = (float)ng[4 + r];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bt = rowt[r] + tc;
      // 16-bit labels are read from ltile at the top of their step; fp32 labels / dY by the owner lane
      s.dy[r] = Lab ? (IO ? 0.f : ((const float*)Labw)[bt * H + jc]) : dYw[bt * H + jc];
      s.r[r] = r4[r];
      s.z[r] = z4[r];
      s.n[r] = n4[r];
      s.ghn[r] = g4[r];
      const float hp = IO ? 0.f : Yw[(bt - (tc > 0 ? 1 : 0)) * H + jc];
      s.hp[r] = tc > 0 ? hp : 0.f;
    }
    if (IO && tc > 0) {                                  // h_{t-1} = 5th component of step t-1's record
      const f32x4 h4 = (gatesw + (size_t)(tc - 1) * NW * GREC * 64)[(GREC - 1) * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r) s.hp[r] = h4[r];
    }
  };
  StepIn cur, nxt;
  load_step(T - 1, cur);
  float ycur[4] = {0.f, 0.f, 0.f, 0.f};
  if (Lab) {
    if (IO) {
      const f32x4 h4 = (gatesw + (size_t)(T - 1) * NW * GREC * 64)[(GREC - 1) * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r) ycur[r] = h4[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) ycur[r] = Yw[(rowt[r] + T - 1) * H + jc];
    }
    if (IO) {
      load_lab(T - 1);
      stage_lab(T - 1);
    }
  }
  f32x4 dhn = {0.f, 0.f, 0.f, 0.f};
  // columns of this lane's unit in the dgh row (r | z | pad | n r) and the dgi row (r | z | n); lanes past H: the dump halfs
  const int o_pad = 32 * KSB + 8 + (c & 7);
  const int o_r = jv ? j : o_pad, o_z = jv ? H + j : o_pad, o_n = jv ? MS + j : o_pad, o_t = jv ? 2 * H + j : o_pad;
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    _Float16* dbuf = dbuf2 + (t & 1) * 4 * MB * DS;
    _Float16* dhi = dbuf;
    _Float16* dlo = dbuf + MB * DS;
    _Float16* ihi = dbuf + 2 * MB * DS;
    _Float16* ilo = dbuf + 3 * MB * DS;
    load_step(t - 1, nxt);
    if (IO && Lab) load_lab(t - 1);
    const float* lt = ltile + (t & 1) * MB * HY;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * g + r;
        const float dyv = Lab ? (ycur[r] - (IO ? lt[m * HY + jc] : cur.dy[r])) * coef : cur.dy[r];
        const float dh = rowok[r] ? dyv * s_in + dhn[r] : 0.f;
        const float rg = cur.r[r], zg = cur.z[r];
        const float ng = X3 ? tanh_fast(__builtin_fmaf(rg, cur.ghn[r], cur.n[r])) : cur.n[r];   // the forward's n, bit for bit
        const float dn = dh * (1.f - zg);
        const float dz = dh * (cur.hp[r] - ng);
        const float dnt = dn * (1.f - ng * ng);
        const float dr = dnt * cur.ghn[r];
        const float dar = dr * rg * (1.f - rg);
        const float daz = dz * zg * (1.f - zg);
        const float dnr = dnt * rg;
        acc[r] = dh * zg;
        // No branch on jv and no read-back: lanes past H aim at the row's 8 dump halfs (never read as K, never copied out),
        // so the four rows' chains sit in one basic block, and dar / daz go to both tiles from registers (the former
        // `ihi[..] = dhi[..]` copies were 16 serialized LDS round trips per step): 101.2 -> 97.5 us (r4, same box)
        const _Float16 arh = (_Float16)dar, azh = (_Float16)daz, nrh = (_Float16)dnr, nth = (_Float16)dnt;
        dhi[m * DS + o_r] = arh;
        dhi[m * DS + o_z] = azh;
        dhi[m * DS + o_n] = nrh;
        ihi[m * DS + o_r] = arh;
        ihi[m * DS + o_z] = azh;
        ihi[m * DS + o_t] = nth;
        if (X3) {
          const _Float16 arl = (_Float16)(dar - (float)arh), azl = (_Float16)(daz - (float)azh);
          dlo[m * DS + o_r] = arl;
          dlo[m * DS + o_z] = azl;
          dlo[m * DS + o_n] = (_Float16)(dnr - (float)nrh);
          ilo[m * DS + o_r] = arl;
          ilo[m * DS + o_z] = azl;
          ilo[m * DS + o_t] = (_Float16)(dnt - (float)nth);
        }
      }
    }
    if (IO && Lab) stage_lab(t - 1);    // other parity than the labels read above; visible after the barrier
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {   // planes out: rows (b, t) of dGI and of dGHn
      if (c_dst[i]) {
        const h8 v = *(const h8*)(dbuf + c_lds[i]);
        *(h8*)(c_dst[i] + (size_t)t * c_step[i]) = v;
      }
    }
    if (active && t > 0) {
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        Frag a;
        a.hi = *(const h8*)(dhi + c * DS + 32 * ks + 8 * g);
        if (X3) a.lo = *(const h8*)(dlo + c * DS + 32 * ks + 8 * g);
        else a.lo = a.hi;
        acc = mfma3<X3>(a, WT[ks], acc);
      }
    }
    dhn = acc;
    __syncthreads();   // measured: without it the waves drift apart and the step gets 8 % slower (LDS contention)
#pragma unroll
    for (int r = 0; r < 4; ++r) ycur[r] = cur.hp[r];   // Y[b, t-1]
    cur = nxt;
  }
}

}  // namespace

bool grux_shape_supported(int H) { return H >= 1 && H <= 127; }

int grux_hp(int H) { return 32 * cdiv_i(H + 1, 32); }
size_t grux_gates_floats(int B, int T, int H, int io) {
  return (size_t)cdiv_i(B, MB) * T * cdiv_i(H, 16) * grec(io) * 64 * 4;
}
int grux_blocks(int B) { return cdiv_i(B, MB); }
// bytes the recurrences really move for the stash: fp16 records in the one-pass mode (grux_gates_floats() sizes the buffer
// for the split modes' fp32 records whatever the mode)
static double gates_bytes(int B, int T, int H, int io, bool x3) {
  return (double)cdiv_i(B, MB) * T * cdiv_i(H, 16) * (x3 ? (io ? 4 : 3) : (io ? 3 : 2)) * 64 * 16.0;
}

int launch_grux_fwd(int B, int T, int H, const void* GI /*fp32 rows (x3) or fp16 rows (one-pass fp16)*/, int ldgi, const float* Whh, const float* bhh, void* Y,
                    float* gates, void* y_planes /*nullable: 2 x [B*T+1][grux_hp(H)] halfs*/, bool x3, unsigned* status,
                    const void* labels /*nullable*/, float* stat_part /*2 * grux_blocks(B) floats if labels*/, int io,
                    int last_only /*Y is [B][H]: only h_{T-1} * y_mul + y_add is written (fp32 I/O, no stash)*/,
                    float y_mul, float y_add, hipStream_t st) {
  if (last_only && (io != 0 || gates || y_planes || labels)) return WGNN_ERR_UNSUPPORTED;
  _Float16* yh = (_Float16*)y_planes;
  _Float16* yl = yh ? yh + ((size_t)B * T + 1) * grux_hp(H) : nullptr;   // each plane has B*T + 1 rows
  const double bt = (double)B * T;
  const double fl = bt * 2.0 * 3 * H * H,
               by = bt * ((x3 ? 4.0 : 2.0) * 3 * H + (io ? 2.0 : 4.0) * ((last_only ? 0 : H) + (labels ? H : 0))) +
                    (gates ? gates_bytes(B, T, H, io, x3) : 0.0);
  const dim3 grid(cdiv_i(B, MB));
#define FLAUNCH(K, X3V, IOV, NAME)                                                                                 \
  PROF_LAUNCH(NAME, fl, by, st,                                                                                    \
              hipLaunchKernelGGL((grux_fwd_kernel<K, X3V, IOV>), grid, dim3(NTHREADS), 0, st, B, T, H, GI, ldgi, Whh, bhh, Y, \
                                 gates, yh, yl, status, labels, stat_part, io, last_only, y_mul, y_add))
#define FCASE(K)                                                                                                   \
  if (x3 && !io) FLAUNCH(K, true, false, "grux_fwd_kernel<" #K ">");                                               \
  else if (x3) FLAUNCH(K, true, true, "grux_fwd_kernel<" #K ">");                                                  \
  else if (!io) FLAUNCH(K, false, false, "grux_fwd_kernel<" #K ",f16>");                                           \
  else FLAUNCH(K, false, true, "grux_fwd_kernel<" #K ",f16>")
  switch (cdiv_i(H + 1, 32)) {
    case 1: FCASE(1); break;
    case 2: FCASE(2); break;
    case 3: FCASE(3); break;
    case 4: FCASE(4); break;
    default: return WGNN_ERR_UNSUPPORTED;
  }
#undef FCASE
#undef FLAUNCH
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}

int grux_hn(int H) { return 8 * cdiv_i(H, 8); }
int grux_msplit(int H) { return 8 * cdiv_i(2 * H, 8); }

int launch_grux_bwd(int B, int T, int H, const float* Whh, const void* Y, const float* dY, const void* labels, int io,
                    const float* gates, const float* GI /*x3: the forward's GI rows [B*T][ldgi] fp32 (stash)*/, int ldgi,
                    const float* scales, void* dGI_planes, void* dGHn_planes, int ldd, bool x3, const float* stat_part,
                    int64_t n_loss, float grad_scale, float* loss, float* scales_out, unsigned* status, int write_lo,
                    hipStream_t st) {
  // stat_part != null: loss and scales are finalised inside the kernel (once a separate one-block launch)
  const int nstat = grux_blocks(B);
  const float inv_n = stat_part ? 1.0f / (float)n_loss : 0.f, coef_in = stat_part ? 2.0f * grad_scale / (float)n_loss : 0.f;
  _Float16* ih = (_Float16*)dGI_planes;
  _Float16* il = ih + (size_t)B * T * ldd;
  _Float16* nh = (_Float16*)dGHn_planes;
  _Float16* nl = nh + (size_t)B * T * grux_hn(H);
  const int ksb = cdiv_i(grux_msplit(H) + H, 32);
  if (ldd % 8 != 0 || ldd < 3 * H || ldd > 32 * ksb) return WGNN_ERR_SHAPE;
  if (x3 && (!GI || ldgi < 3 * H)) return WGNN_ERR_NULL;
  const double bt = (double)B * T;
  const double fl = bt * 2.0 * 3 * H * H,
               by = bt * ((io ? 2.0 + 4.0 : 4.0 + 4.0) * H + (x3 && write_lo ? 4.0 : 2.0) * (3 * H + H)) +
                    gates_bytes(B, T, H, io, x3) + (x3 ? bt * 4.0 * H : 0.0);   // Y + labels in, planes out, stash (+ GI's n third) in
  const dim3 grid(cdiv_i(B, MB));
#define BLAUNCH(K, X3V, IOV, NAME, BYTES)                                                                         \
  PROF_LAUNCH(NAME, fl, BYTES, st,                                                                                \
              hipLaunchKernelGGL((grux_bwd_kernel<K, X3V, IOV>), grid, dim3(NTHREADS), 0, st, B, T, H, Whh, Y, dY, labels, io, \
                                 gates, GI, ldgi, scales, ih, il, ldd, nh, nl, stat_part, nstat, inv_n, coef_in, loss, scales_out, status, write_lo))
#define BCASE(K)                                                                                                  \
  if (x3 && !io) BLAUNCH(K, true, false, "grux_bwd_kernel<" #K ">", by);                                          \
  else if (x3) BLAUNCH(K, true, true, "grux_bwd_kernel<" #K ">", by);                                             \
  else if (!io) BLAUNCH(K, false, false, "grux_bwd_kernel<" #K ",f16>", by);                                      \
  else BLAUNCH(K, false, true, "grux_bwd_kernel<" #K ",f16>", by)
  switch (ksb) {
    case 1: BCASE(1); break;
    case 2: BCASE(2); break;
    case 3: BCASE(3); break;
    case 4: BCASE(4); break;
    case 5: BCASE(5); break;
    case 6: BCASE(6); break;
    case 7: BCASE(7); break;
    case 8: BCASE(8); break;
    case 9: BCASE(9); break;
    case 10: BCASE(10); break;
    case 11: BCASE(11); break;
    case 12: BCASE(12); break;
    default: return WGNN_ERR_UNSUPPORTED;
  }
#undef BCASE
#undef BLAUNCH
  WGNN_CHECK_LAUNCH();
  return WGNN_OK;
}
