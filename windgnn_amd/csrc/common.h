// Shared device helpers for the gfx950 kernels of libwindgnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/windgnn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WGNN_WAVE 64

static inline int cdiv_i(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define WGNN_CHECK_LAUNCH()                              \
  do {                                                   \
    if (hipGetLastError() != hipSuccess) return WGNN_ERR_HIP; \
  } while (0)

// v_mfma_f32_16x16x4_f32: A lane l = A[m = l&15][k = l>>4], B lane l = B[k = l>>4][n = l&15],
// C/D lane l reg r = C[row = 4*(l>>4) + r][col = l&15].  Exact fp32 fmaf chain.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x2_f32: A lane l = A[i = l&31][k = l>>5], B lane l = B[k = l>>5][j = l&31],
// C/D lane l reg r = C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// Hardware-transcendental forms (v_exp_f32 + v_rcp_f32, ~1 ulp each): 4 instructions instead of the
// ~30 of the IEEE-exact expf + division.  |error| < 3e-7, far inside the path's 1e-4 budget; the
// limits are exact (exp2 -> 0 or inf, rcp(inf) = 0).
__device__ __forceinline__ float sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * x) + 1.0f);
}
__device__ __forceinline__ float tanhf_(float x) {
  // 1 - 2/(e^{2x}+1): exact limits at +-inf, abs error ~1e-7
  return 1.0f - 2.0f / (expf(2.0f * x) + 1.0f);
}

// Opt a kernel into > 64 KB of dynamic LDS once per device (hipFuncSetAttribute takes a global lock
// and is far too slow to repeat on every launch).
#include <atomic>
static inline int ensure_dyn_smem(const void* fn, size_t bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return WGNN_ERR_HIP;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return WGNN_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return WGNN_ERR_HIP;
  done.fetch_or(bit, std::memory_order_release);
  return WGNN_OK;
}

// ---- optional per-kernel timing (hipEvents on the launch stream); off by default ----------------
// PROF_LAUNCH(name, flops, bytes, st, launch) wraps one kernel launch; algorithmic flops/bytes are
// what bench.py's roofline uses.  Enabled only through wgnn_profile_enable().
void prof_begin(const char* kernel, double flops, double bytes, hipStream_t st);
void prof_end(hipStream_t st);
extern bool g_prof_on;
#define PROF_LAUNCH(name, flops, bytes, st, ...) \
  do {                                            \
    if (g_prof_on) prof_begin(name, flops, bytes, st); \
    __VA_ARGS__;                                  \
    if (g_prof_on) prof_end(st);                  \
  } while (0)

// ---- range status (include/windgnn.h "Status block"): word 0 of the caller's workspace, only ever OR-ed into.
// The fp16-plane modes cannot represent |x| > 65504 (fp16's largest finite value; anything beyond is flagged); where a kernel converts activations, weights or final
// gradients it tests them (one v_cmp per value, NaN counts as out of range) and reports instead of letting an
// inf/NaN slide through a ReLU (fmaxf(NaN, 0) = 0 would hide it).
#define WGNN_FP16_MAX 65504.0f
__device__ __forceinline__ bool out_of_fp16_range(float v) { return !(__builtin_fabsf(v) <= WGNN_FP16_MAX); }
__device__ __forceinline__ void report_status(unsigned* status, bool bad, unsigned bit) {
  if (status && bad) atomicOr(status, bit);   // call once per thread, after the loop (error path only)
}

// ---- internal launchers (defined in the .hip files, used by api.hip) -------------------------
struct GemmArgs {
  const float* A; int lda; int a_kcontig;   // A(m,k) = a_kcontig ? A[m*lda+k] : A[k*lda+m]
  const float* B; int ldb; int b_kcontig;   // B(k,n) = b_kcontig ? B[n*ldb+k] : B[k*ldb+n]
  float* C; int ldc;                        // C[m*ldc+n]
  int M, N, K;
  const float* bias;                        // optional, added per column n
  int ones_col;                             // if 1, B gets a virtual extra column N-1 of ones (N includes it)
  int shift_T;                              // if >0: B row k reads row k-1, rows with k % shift_T == 0 are zero
  int splitk;                               // >1: partials to `partial` [splitk][M][N], reduce separately
  float* partial;
};
int launch_gemm_f32(const GemmArgs& g, hipStream_t st);
int gemm_f32_nt_splitk(int M, int N, int K);                          // 1 = no split (enough tiles to fill the chip)
int launch_gemm_f32_nt(GemmArgs g, float* kpart, hipStream_t st);     // NT product, split-K + fixed-order sum when few rows
void gemm_f32_tile(int M, int N, int* bm, int* bn);
int gemm_f32_tiles(int M, int N);
// gemm32.hip: big-tile exact-fp32 GEMMs for large B*T (operands padded and 16-byte aligned)
bool gemm32_nt_supported(size_t BT, int Kp_f, int Kp_b);
bool gemm32_tn_supported(size_t BT);
int gemm32_nt_rows(int N);
int launch_pad_weight(const float* W, int R, int C, int transpose, const float* bias, float* out, int Ro, int Co,
                      hipStream_t st);
int launch_gemm32_nt(const float* A, int lda, int M, int Kp, const float* Bp, float* C, int ldc, int N, hipStream_t st);
int gemm32_tn_tiles(int Mo, int No);
int gemm32_tn_pitch(int No);
// A2 != null: GEMM columns m >= msplit of the A operand come from A2 (column m - msplit, row stride lda2)
int launch_gemm32_tn(const float* A, int lda, const float* B, int ldb, int K, int splitk, float* P, int Mo, int No,
                     const float* A2, int lda2, int msplit, hipStream_t st);

int launch_mse_stats(const float* Y, const float* L, int64_t n, float grad_scale, float* loss, float* scales,
                     float* part /*>= 2048 floats*/, hipStream_t st);
int launch_amax_scale(const float* x, int64_t n, float* scales, float* part /*>= 448 floats*/, hipStream_t st);
// split-fp16 GRU recurrences with register-resident weights (grux.hip)
bool grux_shape_supported(int H);
int grux_hp(int H);   // row width (halfs) of the Y planes: 32*ceil((H+1)/32)
size_t grux_gates_floats(int B, int T, int H, int io);   // gate stash of the register-resident recurrences (their own layout)
int launch_grux_fwd(int B, int T, int H, const void* GI /*fp32 rows if x3, else fp16 rows*/, int ldgi, const float* Whh, const float* bhh, void* Y,
                    float* gates, void* y_planes, bool x3, unsigned* status, const void* labels, float* stat_part,
                    int io /*wgnn_io of Y and labels*/, int last_only, float y_mul, float y_add, hipStream_t st);
int grux_blocks(int B);   // workgroups of launch_grux_fwd = MSE partial pairs it writes when given labels
int mse_stats_blocks();
// exactly one of dY / labels is non-null (labels: dY = (Y - labels) * scales[2], see launch_mse_stats)
// dGI planes [B*T][ldd] (3H layout) and the n third of dGH alone, dGHn planes [B*T][grux_hn(H)] (dGH's r and z thirds equal dGI's)
int grux_hn(int H);       // row width (halfs) of the dGHn planes: 8*ceil(H/8)
int grux_msplit(int H);   // 8*ceil(2H/8): first GEMM row of the dGHn block in the dW_hh product
// stat_part != null (the partial pairs + tag wgnn_fwd_loss left in the stash): loss[0], scales_out[0..2] are finalised inside
// the kernel (n_loss = B*T*H, grad_scale as in wgnn_bwd_mse_part); a missing tag gives loss = NaN + WGNN_STATUS_NO_LOSS_STATS
int launch_grux_bwd(int B, int T, int H, const float* Whh, const void* Y, const float* dY, const void* labels, int io,
                    const float* gates, const float* GI /*split modes: the forward's GI rows (stash), n recomputed*/, int ldgi,
                    const float* scales, void* dGI_planes, void* dGHn_planes, int ldd, bool x3, const float* stat_part,
                    int64_t n_loss, float grad_scale, float* loss, float* scales_out, unsigned* status,
                    int write_lo /*0: dGI / dGHn as one fp16 plane (x3 only)*/, hipStream_t st);
#define WGNN_STATS_TAG 20261004.0f   // float word behind the MSE partial pairs: "wgnn_fwd_loss wrote these"

int launch_gcn_partial_reduce(const float* partial, int nblk, float* dW1, float* db1, float* dW2, float* db2,
                              unsigned* status, hipStream_t st);
// register-chained exact-fp32 variants (gcn32.hip): what the WGNN_MATH_F32 path runs
size_t gcn32_bwd_partial_floats(int ntiles);
int launch_gcn32_fwd(int ntiles, int S, const float* A, const float* X, const float* W1, const float* b1,
                     const float* W2, const float* b2, float* g, int ldg, float* xtail_scratch, hipStream_t st);
int launch_gcn32_bwd(int ntiles, int S, const float* A, const float* X, const float* W1, const float* b1,
                     const float* W2, const float* g, int ldg, const float* dg, int ld_dg, float* dW1, float* db1, float* dW2,
                     float* db2, float* partial, float* xtail_scratch, hipStream_t st);
// register-chained split-fp16 variants (gcnx.hip)
size_t gcnx2_bwd_partial_floats(int ntiles);
int gcnx_bwd_grid(int ntiles, int S, bool x3);
// g_planes: fp16 hi plane [ntiles][ldg] followed by the lo plane; column S*13 holds 1.0, later columns 0
int launch_gcnx2_fwd(int ntiles, int S, const float* A, const void* X, int io /*wgnn_io of X*/, const float* W1, const float* b1,
                     const float* W2, const float* b2, void* g_planes, int ldg, bool x3, unsigned* status, void* xtail_scratch,
                     hipStream_t st);   // xtail_scratch: >= S*13 + 1 elements of workspace when S*13 is odd
int launch_gcnx2_bwd(int ntiles, int S, const float* A, const void* X, int io, const float* W1, const float* b1,
                     const float* W2, const void* g_planes, int ldg, const void* dg, int ld_dg /*row pitch of dg, elements*/,
                     bool dg16 /*dg is one fp16 plane*/, const float* scales, int scale_in, float* partial, bool x3,
                     void* xtail_scratch, hipStream_t st);
// fused forward front end (gcngi.hip): both GCN layers + the GRU input projection, g handed over through LDS.
// g_planes (nullable) / stash_planes: the backward's copy of g (0: none, 1: hi plane, 2: hi + lo); Bplanes: the stage-major
// image of [W_ih | b_ih] (launch_split_weight2); GI: fp32 rows (x3) or ONE fp16 plane (one-pass mode), row pitch ldgi elements
bool gcngi_supported(int S, int H, bool x3);
int launch_gcngi_fwd(int ntiles, int S, const float* A, const void* X, int io, const float* W1, const float* b1,
                     const float* W2, const float* b2, void* g_planes, int ldg, int stash_planes, const void* Bplanes,
                     int Np, void* GI, int ldgi, int N, bool x3, unsigned* status, void* xtail_scratch, hipStream_t st,
                     int role_split = 0, int gemm_prio = 0);
// GraphConvLayer with any feature widths, S <= 64 dense (gcn_any.hip), and the g / dg hand-over of wgnn_gru_fwd / wgnn_gru_bwd
bool gcn_any_supported(int S, int Fi, int Fo);
size_t gcn_any_bwd_partial_floats(int ntiles, int Fi, int Fo);
int launch_gcn_any_fwd(int ntiles, int S, int Fi, int Fo, const float* A, const float* X, const float* W, const float* b, float* out,
                       hipStream_t st);
int launch_gcn_any_bwd(int ntiles, int S, int Fi, int Fo, const float* A, const float* X, const float* W, const float* out,
                       const float* dout, float* dW, float* db, float* dX, float* partial, hipStream_t st);
int launch_pack_g(const float* gin, size_t rows, int I, float* g, int ld, hipStream_t st);
int launch_unpack_dg(const float* dg, size_t rows, int I, int ld, float* out, hipStream_t st);
// large-shape NT plane GEMM (pgemm_big.hip); opt_big_gemm(): WGNN_OPT_BIG_GEMM (api.hip), 1 unless switched off for an A/B
bool pgemm_nt256_wanted(int M, int N, int Kp);
int launch_pgemm_nt256(const void* Aimg_hi, const void* Aimg_lo, int M, int k0, int klen, const void* Bplanes, int Np,
                       size_t bplane, float* C, int ldc, int N, bool accumulate, hipStream_t st);
size_t pgemm_nt256_aimg_bytes(int M, int N, int Kp, int planes);
int launch_pgemm_repack_a(const void* Ahi, const void* Alo, int lda, int M, int Kp, void* img, hipStream_t st);
int opt_gemm32_form(); // WGNN_OPT_GEMM32_FORM: 0 one 8-wave workgroup per CU, v >= 1 two 4-wave ones, stagger v - 1 (gemm32.hip)
int opt_big_gemm();   // WGNN_OPT_BIG_GEMM: 0 off, 1 on (needs the caller's A-image scratch)
// ---- The image of a B operand (weights: W_ih | b_ih, W_ih^T, W_hh | b_hh), written by split_weight2_kernel / finish.hip and
// staged by the NT plane GEMMs and the fused front end: one fp16 plane of B[Np][Kp] is STAGE-major (a 32-deep K step of all
// Np rows is contiguous) and, since round 5, FRAGMENT-major inside a stage: the 16 rows x 32 k of one MFMA B fragment are 1 KB
// laid out [k chunk c = (k >> 3) & 3][row r = n & 15][8 halfs], i.e. lane l = 16 c + r of a 16x16x32 B fragment owns bytes
// [16 l, 16 l + 16).  A fragment is then ONE linear 1 KB load (global_load_dwordx4, or LDS-DMA followed by a linear
// ds_read_b128): the row-major form of rounds 1-4 ([n][32 k], 64-byte rows) made the fused kernel's register loads lane-
// transposed (consecutive lanes 64 bytes apart), which the CU's address path serves at 30 B/clk against 51-54 for the linear
// form (tools/l2_stream.hip, profiles/r5_l2_stream.txt).
__host__ __device__ __forceinline__ size_t bimg_off(int n, int k, int Np) {
  return ((size_t)(k >> 5) * (size_t)(Np >> 4) + (size_t)(n >> 4)) * 512 + (size_t)(((k >> 3) & 3) * 128 + (n & 15) * 8 + (k & 7));
}
// plane GEMMs (pgemm.hip)
int launch_split_weight2(const float* W, int R, int C, int transpose, const float* bias, int bias_col, void* planes,
                         int Rp, int Cp, unsigned* status, hipStream_t st);
int pgemm_nt_np(int N);
int pgemm_tn_tiles(int Mout, int Nout);   // output tiles per K chunk (sizes the split-K factor)
size_t pgemm_tn_partial_floats(int Mout, int Nout, int splitk);
// kpart (nullable): pgemm_nt_kpart_floats(M, ldc, Kp) floats of split-K scratch for few-row, long-K products
size_t pgemm_nt_kpart_floats(int M, int ldc, int Kp);
// out16: C is ONE fp16 plane with row pitch ldc halfs (single-plane A operand only: the x2 and f16 instances)
int launch_pgemm_nt(const void* Ahi, const void* Alo, int lda, int M, int Kp, const void* Bplanes, int Np, float* C,
                    int ldc, int N, const float* s_out, bool x3, float* kpart, hipStream_t st, bool out16 = false,
                    void* aimg = nullptr /* pgemm_nt256_aimg_bytes() of scratch: lets the large-shape kernel stage A as an image */);
// A2hi != null: columns m >= msplit of the A operand are column m - msplit of the planes A2hi / A2lo (row stride lda2)
int launch_pgemm_tn(const void* Ahi, const void* Alo, int lda, const void* Bhi, const void* Blo, int ldb, int shift_T,
                    int K, int splitk, float* partial, int Mout, int Nout, bool x3, const void* A2hi, const void* A2lo,
                    int lda2, int msplit, hipStream_t st, bool b_stream = false /*B: old read-once data, non-temporal*/);
// general shapes (general.hip): CSR adjacency (blob layout: see include/windgnn.h) and any hidden width
size_t gcn_csr_bwd_partial_floats();
int launch_gcn2_csr_fwd(int ntiles, int S, int nnz, const void* csr, const float* X, const float* W1, const float* b1,
                        const float* W2, const float* b2, float* h1, float* g, void* g_planes, size_t ldg, bool x3,
                        unsigned* status, hipStream_t st);
int launch_gcn2_csr_bwd(int ntiles, int S, int nnz, const void* csr, const float* X, const float* W2, const float* h1,
                        const float* g, const void* g_hi, size_t ldg, const float* dg, size_t ld_dg,
                        const float* scales, float* du, float* partial, float* dW1, float* db1, float* dW2, float* db2,
                        hipStream_t st);
int launch_gru_gen_fwd_x3(int B, int T, int H, const float* GI, int ldgi, const void* whh_planes, int np_g3,
                          const float* bhh, float* Y, float* gates, void* y_planes, float* gh, float* kpart, void* hc,
                          bool x3, hipStream_t st);
int launch_gru_gen_bwd_x3(int B, int T, int H, const void* whhT_planes, int np_h, const float* Y, const float* dY,
                          const float* gates, const float* scales, void* dgi_planes, void* dgh_planes, int ldd,
                          float* dhz, float* dhw, float* kpart, void* dc, bool x3, hipStream_t st);
int launch_gru_gen_fwd(int B, int T, int H, const float* GI, int ldgi, const float* Whh, const float* bhh, float* Y,
                       float* gates, float* gh, hipStream_t st);
int launch_gru_gen_bwd(int B, int T, int H, const float* Whh, const float* Y, const float* dY, const float* gates,
                       float* dGI, float* dGH, int ldd, float* dhz, float* dhw, hipStream_t st);
size_t gcn1_csr_bwd_ws_floats(int ntiles, int S);
int launch_gcn1_csr_fwd(int ntiles, int S, int nnz, const void* csr, const float* X, const float* W, const float* b,
                        float* out, hipStream_t st);
int launch_gcn1_csr_bwd(int ntiles, int S, int nnz, const void* csr, const float* X, const float* W, const float* out,
                        const float* dout, float* dW, float* db, float* dX, float* ws, hipStream_t st);
int launch_gcn1_fwd(int ntiles, int S, const float* A, const float* X, const float* W,
                    const float* b, float* out, hipStream_t st);
size_t gcn1_bwd_partial_floats(int ntiles);
int launch_gcn1_bwd(int ntiles, int S, const float* A, const float* X, const float* W,
                    const float* out, const float* dout, float* dW, float* db, float* dX,
                    float* partial, hipStream_t st);

// exact-fp32 recurrences with register-resident W_hh (gru.hip), H <= 128
//   forward: gates (nullable) = gru_gates_floats() floats of stash in the kernels' own layout; labels + stat_part (nullable):
//   2 * gru_blocks(B) MSE partials + a tag word; hprev (nullable): [B*T][hq] rows [h_{t-1} | 1 | 0..] for the dW_hh GEMM;
//   last_only: Y is [B][H] and receives only h_{T-1} * y_mul + y_add
int launch_gru_fwd(int B, int T, int H, const float* GI, int ldgi, const float* Whh, const float* bhh, float* Y,
                   float* gates, const float* labels, float* stat_part, float* hprev, int hq, int last_only, float y_mul,
                   float y_add, hipStream_t st);
//   backward: exactly one of dY / labels; dGI [B*T][ldd] and EITHER dGHn [B*T][gru_hn(H)] (dGH's r and z thirds equal
//   dGI's) OR the full dGH [B*T][ldd]; stat_part (nullable, with labels): loss[0] is finalised from the forward's partials
int launch_gru_bwd(int B, int T, int H, const float* Whh, const float* Y, const float* dY, const float* labels,
                   const float* gates, const float* GI /*forward's GI rows (stash): n is recomputed*/, int ldgi, float* dGI, int ldd, float* dGHn, float* dGH, const float* stat_part,
                   int64_t n_loss, float grad_scale, float* loss, unsigned* status, hipStream_t st);
bool gru_shape_supported(int H);
size_t gru_gates_floats(int B, int T, int H);
int gru_blocks(int B);
int gru_hn(int H);
int gru_msplit(int H);
// small batches (B <= 768, H <= 128), exact fp32, one window per workgroup with W_hh in registers (gru_small.hip)
bool gru_small_supported(int B, int H);
int launch_gru_small_fwd(int B, int T, int H, const float* GI, int ldgi, const float* Whh, const float* bhh, float* Y,
                         float* gates, float* hprev /*nullable: [B*T][hq] rows [h_{t-1} | 1 | 0..]*/, int hq,
                         const float* labels /*nullable*/, float* stat_part /*2 * gru_small_blocks(B) + 1 floats*/,
                         hipStream_t st);
int gru_small_blocks(int B);
// exactly one of dY / labels (labels: dY formed in the kernel, loss finalised from stat_part as launch_gru_bwd)
int launch_gru_small_bwd(int B, int T, int H, const float* Whh, const float* Y, const float* dY, const float* labels,
                         const float* gates, float* dGI, float* dGH, int ldd, const float* stat_part, int64_t n_loss,
                         float grad_scale, float* loss, unsigned* status, hipStream_t st);

int launch_mse(const float* Y, const float* L, int64_t n, float scale, float* dY, float* loss,
               float* ws, hipStream_t st);

// ---- finish.hip: reduction of the deferred partial sums + Adam + the prepared W_ih images, one launch
constexpr int TN_BM = 320, TN_WAVES = 8;     // pgemm_tn_kernel's workgroup tile rows / waves (its partial layout)
void pgemm_tn_geom(int Mgemm, int Nout, int* T, int* nNb, int* ntiles);
int gcn32_bwd_grid(int ntiles, int S);       // partial rows launch_gcn32_bwd writes
int gcn_csr_bwd_rows();                      // partial rows launch_gcn2_csr_bwd writes
struct FinSeg {                // one split-K weight-gradient product: C[Mout][ncols] and its bias column Nout - 1
  const float* partial;
  int splitk;
  int kind;                    // 0: nothing to reduce; 1: plain [z][Mout][Nout]; 2: pgemm_tn_kernel's layout
  int T, nNb, ntiles;          // kind 2
  int Mout, Nout, ncols;
  int pitch;                   // kind 1: floats between partial rows (Nout, or the LDS-DMA TN kernel's 64-byte aligned pitch)
  int Mgemm;                   // rows of the GEMM that wrote the partials (= Mout unless its A operand had two sources)
  int msplit, rows1;           // two-source A operand: GEMM row m < rows1 is weight row m, m >= msplit is rows1 + m - msplit
  int scaled;                  // the partials are in units scaled by scales[0]: multiply by scales[1]
  int nblocks;                 // filled in by launch_finish
};
struct FinishArgs {
  FinSeg ih, hh;
  const float* conv_partial;   // [conv_rows][544] per-workgroup partials of the GCN backward, or null
  int conv_rows, conv_blocks, elem_blocks;
  unsigned elem_mask;          // bit t: tensor t's gradient is final in g[t] already (Adam reads it from there)
  const float* scales;
  unsigned* status;
  float* g[8];                 // state_dict order: conv1.w, conv1.b, conv2.w, conv2.b, w_ih, w_hh, b_ih, b_hh
  float* p[8];
  float* m[8];
  float* v[8];
  int n[8];
  int adam;
  float lr_over_bc1, inv_sqrt_bc2, b1, b2, eps;
  int prep_kind;               // 0: none; 1: fp16 hi/lo stage-major planes; 2: zero-padded fp32 copies
  _Float16 *pf_hi, *pf_lo, *pb_hi, *pb_lo;
  int np_g3, np_i;
  float *wp, *wt;
  int Ip, Gp, I;
};
int launch_finish(FinishArgs a, hipStream_t st);
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr,
                float b1, float b2, float eps, hipStream_t st);
