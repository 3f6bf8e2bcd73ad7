/* windgnn.h — C ABI of libwindgnn_hip.so: the MI355X (gfx950) implementation of WindGNN's
 * 2-layer GCN + GRU forward/backward hot path.
 *
 * The reference has no FFI of its own (it is pure Python on torch); what this library replaces is
 * the body of the reference's operator API for the path, and each entry point cites the
 * reference lines it stands in for (paths relative to the reference repo):
 *
 *   wgnn_fwd            GCN_GRU.forward            src/step6_gcn_gru_combined_model.py:13-27
 *                       (GraphConvLayer.forward x2  src/step5_gcn_layer_model.py:13-23, nn.GRU :23)
 *   wgnn_bwd            loss.backward() through it  src/main.py:79
 *   wgnn_gcn_layer_fwd  GraphConvLayer.forward      src/step5_gcn_layer_model.py:13-23
 *   wgnn_gcn_layer_bwd  its autograd backward       src/main.py:79
 *   wgnn_mse_loss_grad  nn.MSELoss()(out, y) + dY   src/main.py:49,72
 *   wgnn_adam_step      torch.optim.Adam.step()     src/main.py:52,80
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / C++ types.
 *   - Every data pointer is DEVICE memory owned by the caller (workspace and stash included);
 *     the library allocates nothing persistent and frees nothing.  Process-wide state is limited to: a
 *     per-kernel "opted into > 64 KB of dynamic LDS on device d" bit set (written once per device, atomically)
 *     and the wgnn_profile_* measurement aid (off by default, not thread-safe).
 *   - All tensors are contiguous, in the reference's layouts; fp32, except X, Y and the labels, which have the element
 *     type wgnn_dims.io names (fp32 by default, fp16 / bf16 with the fp16-plane math modes):
 *       A [S,S] row-major (or CSR, see wgnn_adj_format), X [B,T,S,F] (F fastest), Y [B,T,H], L [B,T,H],
 *       conv*.weight [F,F] (in,out), conv*.bias [F], w_ih [3H, S*F], w_hh [3H,H], b_ih/b_hh [3H],
 *       GRU gate row order r,z,n (torch nn.GRU).
 *   - Calls are asynchronous and ordered on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream).  Re-entrant across streams and devices; safe from any host thread.
 *   - Return value: 0 on success, negative wgnn_status otherwise; never throws, never aborts.
 *     wgnn_strerror() maps a status to text.
 */
#ifndef WINDGNN_H
#define WINDGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WGNN_VERSION 122 /* 0.1.2: wgnn_params.prepared, wgnn_finish, WGNN_BWD_DEFER; 121: WGNN_FINISH_ADAM_GRU / _CONV; 122: wgnn_set_option */

/* Status block: the first 256 bytes of every `workspace` passed to wgnn_fwd / wgnn_bwd* belong to the library as a
 * sticky status area that kernels only ever OR into; word 0 (uint32) holds the bits below.  The caller zeroes the
 * workspace once when it allocates it, may read word 0 whenever it synchronises anyway, and clears it after
 * handling an error.  The fp16-plane math modes (WGNN_MATH_F16X3 / WGNN_MATH_F16) hold activations and weights as
 * fp16 (hi [+ lo]) and cannot represent magnitudes > 65504 (anything beyond is flagged); the reference's fp32 path has no such limit, so
 * instead of producing inf/NaN (or, behind a ReLU, silently 0) the kernels report it here.  WGNN_MATH_F32 never
 * sets a range bit. */
#define WGNN_STATUS_BYTES 256
#define WGNN_STATUS_ACT_RANGE 1u    /* a GCN pre-activation left fp16's range (or was NaN) in the forward */
#define WGNN_STATUS_WEIGHT_RANGE 2u /* a GRU weight / bias left fp16's range */
#define WGNN_STATUS_GRAD_NONFINITE 4u /* a final gradient is inf / NaN */
#define WGNN_STATUS_NO_LOSS_STATS 8u  /* wgnn_bwd_mse_part(part | 8) on a stash whose last forward was not wgnn_fwd_loss:
                                         loss[0] is NaN and the gradients are not to be used (any math mode) */

typedef enum wgnn_status {
  WGNN_OK = 0,
  WGNN_ERR_NULL = -1,        /* a required pointer is NULL */
  WGNN_ERR_SHAPE = -2,       /* a dimension is <= 0 or outside what the kernels support */
  WGNN_ERR_DTYPE = -3,       /* unsupported dtype / math mode */
  WGNN_ERR_WORKSPACE = -4,   /* workspace or stash smaller than wgnn_*_bytes() says */
  WGNN_ERR_UNSUPPORTED = -5, /* e.g. a dense adjacency with S > 64 (pass it as CSR) */
  WGNN_ERR_HIP = -6,         /* a HIP runtime call or kernel launch failed */
  WGNN_ERR_RANGE = -7        /* host bindings raise this when they read a non-zero status word (see Status block) */
} wgnn_status;

/* math mode of the contractions (the element type of X / Y / labels is wgnn_dims.io, see wgnn_io) */
typedef enum wgnn_math {
  WGNN_MATH_F32 = 0,   /* fp32-input MFMA: bitwise an fp32 fmaf chain */
  WGNN_MATH_F16X3 = 1, /* split-fp16 (hi+lo) MFMA, 3 products, fp32 accumulate: fp32-grade error */
  WGNN_MATH_F16 = 2,   /* plain fp16 operands, one MFMA pass, fp32 accumulate: ~1e-3 error (16-bit config); with the
                          register-resident recurrence (H <= 127) the intermediates GI and dg, too, are single fp16 planes */
  WGNN_MATH_F16X3G = 3 /* F16X3, except that from B*T >= 4096 rows the backward's gate gradients (dGI, dGH_n) leave the BPTT
                          kernel as ONE fp16 plane and the three GEMMs they feed run fewer MFMA passes: dW_hh and dg two
                          (hi x (hi + lo)), dW_ih one (hi x hi); dg itself, too, is written as one fp16 plane (dense
                          adjacency).  Forward, recurrences and the GCN backward's own chain are F16X3's (Y identical).  Gradients: every dropped lo half is a relative rounding of 2^-12, independent
                          per element, that averages out over the B*T rows a weight gradient sums -- observed <= 9e-6 of
                          the tensor's max against the fp64 oracle at B*T = 6144 with the MSE loss (F16X3: 1.3e-6), and
                          at worst ~4e-4 when dY is pure zero-mean noise (a gradient that is itself a fully cancelling
                          sum); 2.0e-6 at B*T = 98304 (F16X3: 9.6e-7).  Below 4096 rows it IS F16X3, bit for bit.
                          With a hidden state wider than the register-resident recurrence (H > 127: the per-step GEMM
                          recurrence) the planes are still written in full and the three GEMMs leave the lo plane of dGI /
                          dGH unread from B*T >= 3072 rows (dW_ih one pass, dW_hh and dg two); below that F16X3. */
} wgnn_math;

/* Adjacency argument `A` of wgnn_fwd / wgnn_bwd:
 *   WGNN_ADJ_DENSE  A [S,S] fp32 row-major, S <= 64 (held in LDS);
 *   WGNN_ADJ_CSR    any S: `A` points to ONE device buffer of 2*(S+1+2*nnz) 32-bit words holding the matrix
 *                   and its transpose (the backward multiplies by A^T), each as
 *                     rowptr int32[S+1] | col int32[nnz] (ascending inside a row) | val fp32[nnz]:
 *                   words [0, S+1+2*nnz) describe A, the next S+1+2*nnz words describe A^T
 *                   (windgnn_amd.graph.CsrAdjacency builds it).  The sparse aggregation itself is always
 *                   exact fp32; `math` selects the kernels of the dense GRU contractions as usual. */
typedef enum wgnn_adj_format { WGNN_ADJ_DENSE = 0, WGNN_ADJ_CSR = 1 } wgnn_adj_format;

/* Element type of the tensors "on the wire": X [B,T,S,F], Y [B,T,H] and the labels of wgnn_fwd_loss / wgnn_bwd_mse_part.
 * WGNN_IO_F32 (default; what every other comment in this header assumes), or one of the 16-bit types for BASELINE's
 * 16-bit configuration (half the algorithmic bytes of the path: 26 112 B per window forward instead of 52 224).
 * Parameters, gradients, dY, the adjacency, workspace and stash are unaffected.  16-bit I/O needs the fp16-plane
 * kernels: math F16X3 or F16, dense adjacency, H <= 127 -- anything else returns WGNN_ERR_UNSUPPORTED.  Y is the
 * fp32 result rounded once on the way out; the loss statistics and the backward use the unrounded hidden state. */
typedef enum wgnn_io { WGNN_IO_F32 = 0, WGNN_IO_F16 = 1, WGNN_IO_BF16 = 2 } wgnn_io;

typedef struct wgnn_dims {
  int32_t B;          /* windows in this call */
  int32_t T;          /* timesteps per window (seq_len) */
  int32_t S;          /* stations (graph nodes) */
  int32_t F;          /* features per node; the reference hard-codes 13 */
  int32_t H;          /* GRU hidden width (reference: 3*S) */
  int32_t math;       /* wgnn_math */
  int32_t adj_format; /* wgnn_adj_format */
  int32_t nnz;        /* CSR only: stored entries of A */
  int32_t io;         /* wgnn_io: element type of X, Y and the labels */
} wgnn_dims;

/* The 8 tensors of the reference state_dict, in its key order. */
typedef struct wgnn_params {
  const float* conv1_weight; /* conv1.weight [F,F] */
  const float* conv1_bias;   /* conv1.bias   [F]   */
  const float* conv2_weight; /* conv2.weight [F,F] */
  const float* conv2_bias;   /* conv2.bias   [F]   */
  const float* w_ih;         /* gru.weight_ih_l0 [3H, S*F] */
  const float* w_hh;         /* gru.weight_hh_l0 [3H, H]   */
  const float* b_ih;         /* gru.bias_ih_l0   [3H]      */
  const float* b_hh;         /* gru.bias_hh_l0   [3H]      */
  /* Optional (NULL = none): a caller-kept device buffer of wgnn_prepared_bytes() bytes holding W_ih in the form the
   * GEMM kernels stage it (fp16 hi/lo stage-major planes of [W_ih | b_ih] and of W_ih^T in the fp16-plane modes,
   * zero-padded fp32 copies in WGNN_MATH_F32).  With NULL, wgnn_fwd* / wgnn_bwd* rebuild those images inside the
   * workspace on every call (two launch-sized passes per call).  A caller that keeps the buffer must refresh it with
   * wgnn_prepare_weights() whenever it changes w_ih / b_ih itself; wgnn_finish(adam) keeps it current. */
  void* prepared;
} wgnn_params;

typedef struct wgnn_grads {
  float* conv1_weight;
  float* conv1_bias;
  float* conv2_weight;
  float* conv2_bias;
  float* w_ih;
  float* w_hh;
  float* b_ih;
  float* b_hh;
} wgnn_grads;

/* torch.optim.Adam state and hyper-parameters for the 8 tensors (src/main.py:52,80). */
typedef struct wgnn_adam {
  wgnn_grads exp_avg;    /* first moments, same 8 slots as the parameters */
  wgnn_grads exp_avg_sq; /* second moments */
  int32_t step;          /* 1-based count of THIS update (bias correction uses it) */
  float lr, beta1, beta2, eps;
} wgnn_adam;

int wgnn_version(void);
const char* wgnn_strerror(int status);

/* Process-wide options: which of two bit-identical kernel schedules runs.  Options 0-3 do not change a result bit, option 4 only the summation order of some products; there is no
 * reference counterpart (the reference has no kernels to choose between).  wgnn_set_option returns the PREVIOUS value (>= 0)
 * or WGNN_ERR_SHAPE for an unknown key / value; it takes effect for calls issued after it returns and is atomic, but callers
 * that flip an option while other threads launch get either schedule for those launches.
 *   WGNN_OPT_FUSED_FWD  0 never / 1 stash-less forwards (default) / 2 every supported forward run the fused GCN + input
 *                       projection kernel (csrc/gcngi.hip).  Initial value: environment variable WGNN_FUSED_FWD, read once at
 *                       the first call that needs it (never again). */
#define WGNN_OPT_FUSED_FWD 0
/* Measurement aids (same results, another schedule; defaults 0 / 0 / 1): roles of the fused kernel's waves assigned per SIMD
 * instead of per wave index (0 / 1); s_setprio level of its projection waves (0..3); backward part 2 (dg GEMM -> GCN backward)
 * as 1 / 2 / 4 / 8 producer -> consumer pairs over row chunks (taken only where every chunk still fills the chip; must not
 * change between a WGNN_BWD_DEFER part 2 and its wgnn_finish). */
#define WGNN_OPT_GG_ROLE_SPLIT 1
#define WGNN_OPT_GG_GEMM_PRIO 2
#define WGNN_OPT_BWD2_CHUNKS 3
#define WGNN_BWD2_MAX_CHUNKS 8
/* 1 (default): NT plane products with >= 1024 rows, >= 2048 columns and a contraction >= 1024 long (BASELINE configs[4]) run
 * the 256 x 256-tile kernel of csrc/pgemm_big.hip (their activation operand rewritten as an image first); 0: the 192 x 448-tile
 * kernel shaped for the 34-station widths.  The two sum each dot product in a different order (results differ by fp32
 * rounding, inside every stated tolerance). */
#define WGNN_OPT_BIG_GEMM 4
/* Exact-fp32 NT products (GI, dg) from 24 576 rows on: 0 one 8-wave workgroup per CU with a 128 x 64 T tile; v >= 1 two 4-wave
 * workgroups per CU with 128 x 32 T tiles, the second of each pair started (v - 1) x 3.4 us late so that one's epilogue falls
 * into the other's K loop (v <= 33); 34 the persistent form of the
 * 8-wave kernel (cross-tile prefetch, counted waits).  Same products, same summation order per element: results are bit-identical. */
#define WGNN_OPT_GEMM32_FORM 5
#define WGNN_OPT_COUNT 6
int wgnn_set_option(int key, int value);
int wgnn_get_option(int key);

/* Bytes of scratch wgnn_fwd / wgnn_bwd need (the larger of the two), and of the forward->backward
 * stash.  Both depend on dims only. */
size_t wgnn_workspace_bytes(const wgnn_dims* d);
size_t wgnn_stash_bytes(const wgnn_dims* d);

/* Y[B,T,H] = GRU(relu(A relu(A X W1 + b1) W2 + b2)), h0 = 0 per window.
 * stash may be NULL (inference: src/main.py:100-102); otherwise it receives what wgnn_bwd needs.
 * In the fp16-plane math modes with a dense adjacency and H <= 127 a stash-less call (and wgnn_fwd_last) runs both graph
 * convolutions and the GRU input projection as ONE kernel whose intermediate g never leaves the chip (csrc/gcngi.hip);
 * WGNN_OPT_FUSED_FWD (above) selects which forwards do.  The results do not depend on it (bit-identical either way). */
int wgnn_fwd(const wgnn_dims* d, const float* A, const void* X /* d->io */, const wgnn_params* p, void* Y /* d->io */,
             void* stash, void* workspace, size_t workspace_bytes, void* stream);

/* wgnn_fwd for a training step whose loss is nn.MSELoss()(Y, labels) (src/main.py:66 + :72): the same Y and stash,
 * and where the forward recurrence holds every h_t in registers anyway (every math mode with H <= 127 / 128; not the
 * wide-GRU path) it also reduces (Y - labels) to per-workgroup partial sums of squares and maxima inside the stash and
 * tags them, so that wgnn_bwd_mse_part(..., part | 8) needs no pass over Y and the labels for the loss, the range scale
 * and dY.  On the wide-GRU path it is exactly wgnn_fwd (and the backward ignores bit 8).  stash must not be NULL. */
int wgnn_fwd_loss(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p, const void* labels,
                  void* Y, void* stash, void* workspace, size_t workspace_bytes, void* stream);

/* N4, the evaluation loop's read-out (src/main.py:100-103,116,131,146) as ONE call: forward without a stash, and
 * out[b][j] = Y[b][T-1][j] * (wind_max - wind_min) + wind_min, [B, H] fp32.  With the register-resident recurrence
 * (f16x3 / f16, H <= 127) the other T-1 rows of Y are never written; the other recurrences write Y into the workspace
 * and read the last row out.  fp32 I/O only. */
int wgnn_fwd_last(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p, float wind_min,
                  float wind_max, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Gradients of sum(Y * dY) w.r.t. the 8 parameters (overwritten, not accumulated).
 * No dX and no dA: neither requires grad in the reference (src/main.py:26,
 * src/step4_sequence_preparer.py:58). */
int wgnn_bwd(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p,
             const void* Y, const float* dY /* always fp32 */, const void* stash, const wgnn_grads* g,
             void* workspace, size_t workspace_bytes, void* stream);

/* The same backward in parts, for callers that overlap them (no reference counterpart: it has no
 * streams or collectives).  `part` is a bit mask, wgnn_bwd == 7:
 *   1  BPTT recurrence (reads dY; writes the dGI / dGH buffers in the workspace);
 *   4  weight-gradient GEMMs: dW_ih, db_ih, dW_hh, db_hh (99.8 % of the gradient bytes) final on return;
 *   2  dg GEMM + GCN backward: the four conv gradients final on return.
 * Parts 4 and 2 only read what part 1 wrote and use disjoint scratch, so after part 1 they may run
 * concurrently on two streams (ordered after part 1 by events), and a data-parallel caller can start
 * all-reducing the GRU gradients as soon as part 4 is done. */
int wgnn_bwd_part(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p,
                  const void* Y, const float* dY, const void* stash, const wgnn_grads* g,
                  void* workspace, size_t workspace_bytes, void* stream, int part /* bit mask 1..7, + WGNN_BWD_DEFER */);

/* The reference's loss call folded into the backward (src/main.py:72 + :79): gradients of
 * grad_scale * mean((Y - labels)^2) w.r.t. the 8 parameters, and loss[0] = mean((Y - labels)^2) (written by the call
 * that has part bit 1).  Same parts, scratch and ordering as wgnn_bwd_part; equal to wgnn_mse_loss_grad followed by
 * wgnn_bwd_part, but dY [B,T,H] is not written or re-read where the recurrence kernel can form it from Y and the
 * labels (f16x3 / f16, H <= 127); other shapes build dY inside the workspace. */
int wgnn_bwd_mse_part(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p,
                      const void* Y, const void* labels /* [B,T,H], d->io */, float grad_scale, float* loss,
                      const void* stash, const wgnn_grads* g, void* workspace, size_t workspace_bytes,
                      void* stream, int part /* bit mask 1..7, + 8: the forward was wgnn_fwd_loss on these labels, + WGNN_BWD_DEFER */);

/* `part` bit of wgnn_bwd_part / wgnn_bwd_mse_part: leave the split-K partial sums of the weight-gradient GEMMs (part 4)
 * and the per-workgroup partial sums of the GCN backward (part 2) un-reduced inside the workspace; `g` is not written.
 * wgnn_finish() on the SAME workspace, before any other call that uses it, reduces them. */
#define WGNN_BWD_DEFER 16

/* The tail of a training step as ONE launch (src/main.py:79-80: the end of loss.backward() + optimizer.step()):
 *   which & 4   reduce the deferred partials of part 4 into g->w_ih, b_ih, w_hh, b_hh (fixed order: deterministic);
 *   which & 2   reduce the deferred partials of part 2 into the four conv gradients;
 *   adam        (NULL = none) torch.optim.Adam semantics on all 8 parameters `p` (updated in place) with the gradients
 *               just reduced, or as they stand in `g` for the parts not named in `which`; if p->prepared is set, the
 *               staged images of the NEW w_ih / b_ih are written too, so the next wgnn_fwd* / wgnn_bwd* need no
 *               re-split / re-pad pass.
 *   which & WGNN_FINISH_ADAM_GRU / WGNN_FINISH_ADAM_CONV (with adam, and with no reduce bit): the optimiser step of the four
 *               GRU tensors (+ the staged images) only / of the four conv tensors only, so that a data-parallel caller can
 *               update the GRU tensors while the conv gradients' all-reduce is still in flight.
 * One rank: wgnn_bwd_mse_part(.., 7 | 8 | WGNN_BWD_DEFER), wgnn_finish(.., 6, &adam).  Data parallel: part 1 | 4 | DEFER,
 * wgnn_finish(4, NULL), all-reduce of the GRU gradients overlapped with part 2 | DEFER, wgnn_finish(2, NULL), all-reduce
 * of the conv gradients overlapped with wgnn_finish(WGNN_FINISH_ADAM_GRU, &adam), wgnn_finish(WGNN_FINISH_ADAM_CONV, &adam)
 * (or one wgnn_finish(0, &adam) after both).  The fused and the split forms give bitwise identical results. */
#define WGNN_FINISH_ADAM_GRU 16
#define WGNN_FINISH_ADAM_CONV 32
int wgnn_finish(const wgnn_dims* d, const wgnn_params* p, const wgnn_grads* g, int which, const wgnn_adam* adam,
                void* workspace, size_t workspace_bytes, void* stream);

/* Bytes of the caller-kept W_ih images (wgnn_params.prepared); depends on S, H and math only; 0 = this configuration
 * stages W_ih as it is (then wgnn_prepare_weights returns WGNN_ERR_UNSUPPORTED and `prepared` is ignored). */
size_t wgnn_prepared_bytes(const wgnn_dims* d);
/* Build the images of p->w_ih / p->b_ih in p->prepared (all of it, padding included).  workspace: the status block. */
int wgnn_prepare_weights(const wgnn_dims* d, const wgnn_params* p, void* workspace, size_t workspace_bytes,
                         void* stream);

/* One GraphConvLayer(F, F_out) (src/step5_gcn_layer_model.py:5-23: weight [F, F_out], bias [F_out]):
 * out[n,S,F_out] = relu(A X[n] W + b) for n = 0..ntiles-1 (ntiles = prod of the leading dims of attr_matrix), dense A with
 * S <= 64, 1 <= F, F_out <= 64.  Backward: dW [F, F_out], db [F_out] (overwritten) and, if dX != NULL, dX [n,S,F].
 * F == F_out == 13 -- the only widths the reference's own model builds (src/main.py:41) -- run the MFMA kernels; other widths
 * an exact-fp32 kernel of their own (csrc/gcn_any.hip).  Wider layers / more stations: WGNN_ERR_UNSUPPORTED. */
size_t wgnn_gcn_layer_workspace_bytes(int32_t ntiles, int32_t S, int32_t F, int32_t F_out);
int wgnn_gcn_layer_fwd(int32_t ntiles, int32_t S, int32_t F, int32_t F_out, const float* A, const float* X,
                       const float* W, const float* b, float* out, void* stream);
int wgnn_gcn_layer_bwd(int32_t ntiles, int32_t S, int32_t F, int32_t F_out, const float* A, const float* X,
                       const float* W, const float* out, const float* dout, float* dW, float* db,
                       float* dX, void* workspace, size_t workspace_bytes, void* stream);

/* The recurrent half of GCN_GRU.forward alone (src/step6_gcn_gru_combined_model.py:23: nn.GRU(gru_input, gru_hidden_dim,
 * batch_first=True), h0 = 0 per window) on a caller-supplied g [B,T,S*13] -- what a GCN_GRU whose input_dim / hidden_dim are
 * not 13 runs behind two wgnn_gcn_layer_* calls (the fused path's kernels hard-code 13 features, like the reference's
 * :16).  d: the same dims as for wgnn_fwd with math WGNN_MATH_F32, io WGNN_IO_F32, a dense adjacency format (else
 * WGNN_ERR_UNSUPPORTED); only p->w_ih, w_hh, b_ih, b_hh are read.  Workspace / stash sizes: wgnn_workspace_bytes /
 * wgnn_stash_bytes of d.  wgnn_gru_bwd writes the four GRU slots of `grads` (the conv slots may be NULL) and dg [B,T,S*13]. */
int wgnn_gru_fwd(const wgnn_dims* d, const float* g, const wgnn_params* p, void* Y, void* stash, void* workspace,
                 size_t workspace_bytes, void* stream);
int wgnn_gru_bwd(const wgnn_dims* d, const float* g, const wgnn_params* p, const void* Y, const float* dY, const void* stash,
                 const wgnn_grads* grads, float* dg, void* workspace, size_t workspace_bytes, void* stream);

/* The same layer over a CSR adjacency (`csr`: the WGNN_ADJ_CSR buffer described at wgnn_adj_format), any S. */
size_t wgnn_gcn_layer_csr_workspace_bytes(int32_t ntiles, int32_t S, int32_t F);
int wgnn_gcn_layer_csr_fwd(int32_t ntiles, int32_t S, int32_t F, int32_t nnz, const void* csr, const float* X,
                           const float* W, const float* b, float* out, void* stream);
int wgnn_gcn_layer_csr_bwd(int32_t ntiles, int32_t S, int32_t F, int32_t nnz, const void* csr, const float* X,
                           const float* W, const float* out, const float* dout, float* dW, float* db,
                           float* dX, void* workspace, size_t workspace_bytes, void* stream);

/* loss[0] = mean((Y-L)^2) over n elements; dY = 2 (Y-L) * grad_scale / n.
 * (grad_scale = 1/world_size under data parallel so that summed shard grads equal the
 * big-batch gradient.)  workspace: >= 4096 bytes. */
int wgnn_mse_loss_grad(const float* Y, const float* L, int64_t n, float grad_scale, float* dY,
                       float* loss, void* workspace, size_t workspace_bytes, void* stream);

/* torch.optim.Adam defaults semantics on flat fp32 buffers of n elements; `step` is the 1-based
 * step count AFTER this update (bias correction uses it). */
int wgnn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   int32_t step, float lr, float beta1, float beta2, float eps, void* stream);

/* N2 (src/step4_sequence_preparer.py:7-21): gather B windows of seq_len steps out of a device-resident
 * feature array feat[Ttot][S][F] (the 13 feature columns, i.e. the reference's columns 2:15):
 *   X[b][t][s][f]   = feat[t0_b + t][s][f]
 *   L[b][t][k*S+s]  = feat[t0_b + t + k + 1][s][label_feat],  k = 0,1,2   (+1/+2/+3 h labels; the
 *                     reference's label column 13 is feature index 11)
 * t0_b = starts[b] (device int32 array; the same values on the host in starts_host for validation), or
 * b*seq_len when both are NULL (the reference's non-overlapping windows, :10-13).  Every window needs
 * t0 + seq_len + 3 <= Ttot, else WGNN_ERR_SHAPE (the reference would build a ragged array there). */
int wgnn_make_windows(const float* feat, int64_t Ttot, int32_t S, int32_t F, int32_t seq_len,
                      int32_t label_feat, const int32_t* starts_host, const int32_t* starts_dev,
                      int32_t B, float* X, float* L, void* stream);

/* N4 (src/main.py:103,116,131,146): out[b][j] = Y[b][T-1][j] * (wind_max - wind_min) + wind_min. */
int wgnn_predict_last(const float* Y, int32_t B, int32_t T, int32_t H, float wind_min, float wind_max,
                      float* out, void* stream);

/* Measurement aid for bench.py (no reference counterpart): while enabled, every kernel launch made
 * by this library is bracketed by hipEvents on its stream and tallied per kernel symbol together
 * with its algorithmic flops/bytes.  wgnn_profile_read(idx, ...) synchronises the device and
 * returns the idx-th kernel's totals, or WGNN_ERR_SHAPE past the end.  Off by default; the only
 * global state in the library; not thread-safe. */
int wgnn_profile_enable(int on);
int wgnn_profile_read(int idx, char* name, size_t name_len, double* total_ms, int64_t* launches,
                      double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* WINDGNN_H */
