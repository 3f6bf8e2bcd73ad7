// extern "C" entry points of libwindgnn_hip.so (declared in include/windgnn.h).
// Orchestrates the kernels of gcn.hip / gemm.hip / gru.hip / train_ops.hip on the caller's stream,
// inside caller-owned workspace and stash buffers.  No allocation, no host synchronisation.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "common.h"

namespace {

inline size_t rup(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Layout {
  size_t BT, I, Ip, G3, Gp, H, Hp;
  size_t Id;                       // row pitch of dg: I rounded up to 16 floats, so that the NT GEMMs' 64-byte store segments are aligned
  int np_g3, np_i;                 // padded plane rows of the two split-weight images (f16x3)
  // forward workspace (float offsets)
  size_t ws_GI, ws_g, ws_planes_f, ws_Ylast, ws_xtail_f, ws_xtail_b, fwd_floats;
  // stash
  size_t st_g, st_gates, st_yp, stash_floats;
  // backward workspace
  size_t ws_dGI, ws_dGH, ws_dg, ws_part_ih, ws_part_hh, ws_gcnpart, ws_planes_b, ws_scales, ws_dY, bwd_floats;
  // caller-kept images of W_ih (wgnn_params.prepared): float offsets inside that buffer, 0 floats = this mode has none
  size_t prep_f, prep_b, prep_floats;
  int prep_kind;                                 // 0 none, 1 fp16 planes, 2 padded fp32
  int sk_ih, sk_hh;
  // which kernels run: the fp16-plane family needs the dense LDS-resident GCN and the register-resident GRU;
  // shapes beyond the fast kernels (CSR adjacency, wide hidden state) use general.hip in exact fp32
  bool x3, gen_gcn, gen_gru, g32, g32tn;
  int hq;
  int np_h;                                      // padded rows of split(W_hh^T) (general f16x3 GRU)
  bool dg16;                                     // ... and dg itself ONE fp16 plane (dense GCN only)
  bool gi16;                                     // one-pass fp16 mode: the input projection GI is ONE fp16 plane
  bool dgi1;                                     // f16x3 at large B*T: dGI / dGHn are ONE fp16 plane, their three GEMMs run two passes
  bool gen2p;                                    // f16x3g on the wide-GRU path: the three GEMMs skip the lo plane of dGI / dGH
  bool small, rec32;                             // exact fp32: one-window-per-workgroup recurrences / the register-resident MFMA ones
  size_t st_hprev;                               // exact fp32, large B*T: [Hprev | 1 | 0..] rows written by the forward recurrence
  bool dghn;                                     // only the n third of dGH is stored (dGHn): fast f16x3 recurrence; rec32 at large B*T
  int hn, msplit, m_hh;                          // its row width, its first GEMM row, GEMM rows of the dW_hh product
  size_t st_h1;                                  // general GCN: layer-1 activations
  size_t st_stats;                               // wgnn_fwd_loss: MSE partial pairs (sum | max) of the forward recurrence
  size_t st_GI;                                  // split modes, register-resident recurrence: GI lives in the stash (BPTT recomputes n from its n third)
  bool gi_stash;
  size_t ws_gh, ws_h1, ws_yp, ws_hhp_f, ws_kp_f, ws_hc, ws_du, ws_dhz, ws_dhw, ws_hhp_b, ws_kp_b, ws_dc;   // general GRU / GCN scratch
  size_t ws_ntk_f, ws_ntk_b;                     // exact fp32, few rows: split-K partial sums of the GI / dg products
  size_t ws_aimg_f, ws_aimg_b;                   // large-shape NT plane GEMMs (configs[4]): the A operand (g / dGI) rewritten as an image
};

// (min_rows = 64 measured against 128 / 192 / 256 at B*T = 6144 in round 3: fewer, longer K chunks cost the split-K GEMMs more
// -- 54 -> 78 -> 103 -> 129 us for the two -- than the smaller partial sums save the finish kernel, 16.7 -> 12.2 -> 10.9 us)
int pick_splitk(size_t BT, int tiles, int target_wgs, int min_rows) {
  int by_rows = (int)(BT / (size_t)min_rows);
  int by_grid = target_wgs / (tiles > 0 ? tiles : 1);
  int sk = by_rows < by_grid ? by_rows : by_grid;
  if (sk >= 8) sk -= sk % 8;       // multiples of 8 keep the tiles of one K chunk on one XCD
  return sk < 1 ? 1 : sk;
}

Layout make_layout(const wgnn_dims* d) {
  Layout L;
  L.x3 = d->math != WGNN_MATH_F32;                                      // the fp16-plane kernel family
  L.gen_gcn = d->adj_format == WGNN_ADJ_CSR;
  L.gen_gru = L.x3 ? !grux_shape_supported(d->H) : !gru_shape_supported(d->H);
  const bool x3 = L.x3;
  L.BT = (size_t)d->B * d->T;
  L.I = (size_t)d->S * d->F;
  L.Ip = rup(L.I + 1, 32);                  // room for the ones column at index I
  L.Id = rup(L.I, 16);                      // (dg's 1768-byte rows made the dg GEMM write 23 % more than its output: 215 vs 174 MB)
  L.H = d->H;
  L.Hp = x3 ? (size_t)grux_hp(d->H) : 0;
  L.G3 = 3 * (size_t)d->H;
  L.Gp = rup(L.G3, 32);
  L.np_g3 = pgemm_nt_np((int)L.G3);
  L.np_i = pgemm_nt_np((int)L.I);
  auto al = [](size_t x) { return align_up(x, 64); };
  // exact fp32 at large B*T: the big-tile GEMMs of gemm32.hip on zero-padded copies of W_ih / W_ih^T
  L.g32 = !x3 && !L.gen_gcn && !L.gen_gru && gemm32_nt_supported(L.BT, (int)L.Ip, (int)L.Gp);
  L.g32tn = L.g32 && gemm32_tn_supported(L.BT);          // the split-K dW products (same threshold today)
  L.small = !x3 && !L.gen_gru && gru_small_supported(d->B, d->H);
  L.rec32 = !x3 && !L.gen_gru && !L.small;
  // the images of W_ih the GEMMs stage: sized from S and H alone (the exact-fp32 ones are USED from B*T >= 4096 only),
  // so that a caller-kept copy (wgnn_params.prepared) serves every batch size
  const bool g32_shape = !x3 && !L.gen_gcn && !L.gen_gru && gemm32_nt_supported(1u << 30, (int)L.Ip, (int)L.Gp);
  const size_t planes_f = x3 ? (size_t)L.np_g3 * L.Ip : (g32_shape ? (size_t)gemm32_nt_rows((int)L.G3) * L.Ip : 0);   // 2 planes of halfs = that many floats
  const size_t planes_b = x3 ? (size_t)L.np_i * L.Gp : (g32_shape ? (size_t)gemm32_nt_rows((int)L.I) * L.Gp : 0);
  L.prep_kind = x3 ? 1 : (g32_shape ? 2 : 0);
  L.prep_f = 0;
  L.prep_b = al(planes_f);
  L.prep_floats = L.prep_kind ? al(planes_f) + al(planes_b) : 0;
  constexpr size_t HDR = WGNN_STATUS_BYTES / sizeof(float);   // status block at the start of the workspace
  size_t o = HDR;
  L.ws_GI = o; o += al(L.BT * L.Gp);   // rows padded to 128-B multiples
  L.ws_g = o; o += al(L.BT * L.Ip);    // fp32 g (f32 mode) or its two fp16 planes (f16x3): same bytes
  L.ws_planes_f = o; o += al(planes_f);
  L.np_h = pgemm_nt_np((int)L.H);
  L.ws_gh = o; o += al(L.gen_gru ? (size_t)d->B * L.Gp : 0);
  L.ws_h1 = o; o += al(L.gen_gcn ? L.BT * L.I : 0);
  L.ws_yp = o; o += al(L.gen_gru && x3 ? (L.BT + 1) * L.Hp : 0);            // h planes when there is no stash
  L.ws_hhp_f = o; o += al(L.gen_gru && x3 ? (size_t)L.np_g3 * L.Hp : 0);    // split(W_hh | b_hh)
  L.ws_kp_f = o; o += al(L.gen_gru && x3 ? pgemm_nt_kpart_floats(d->B, (int)L.Gp, (int)L.Hp) : 0);
  L.ws_hc = o; o += al(L.gen_gru && x3 ? (size_t)d->B * L.Hp : 0);          // compact planes of h_{t-1}
  L.ws_xtail_f = o; o += al((L.I & 1) && !L.gen_gcn ? L.I + 1 : 0);        // private copy of X's last tile (odd S*13: see xtail_copy)
  // wgnn_fwd_last where the recurrence writes all of Y: wgnn_fwd_last has no stash, so g lives in ws_g and is dead once GI
  // is formed -- Y aliases it whenever it fits (H <= Ip: always when H = 3S), and only otherwise gets a region of its own
  if ((x3 && !L.gen_gru) || L.rec32 || L.H <= L.Ip) { L.ws_Ylast = L.ws_g; }
  else { L.ws_Ylast = o; o += al(L.BT * L.H); }
  // the A operand of the large-shape projection GEMM as an image (pgemm_big.hip): g's two planes, when the shape is one of its
  L.ws_aimg_f = o; o += al(x3 ? pgemm_nt256_aimg_bytes((int)L.BT, (int)L.G3, (int)L.Ip, 2) / 4 : 0);
  {
    const int sk = (!x3 && !L.g32 && L.BT < 65536) ? gemm_f32_nt_splitk((int)L.BT, (int)L.G3, (int)L.I) : 1;
    L.ws_ntk_f = o; o += al(sk > 1 ? (size_t)sk * L.BT * L.G3 : 0);
  }
  L.fwd_floats = o;
  o = 0;
  L.st_g = o; o += al(L.BT * L.Ip);
  {  // r, z, n, gh_n of every step: [B*T][4H] fp32, or the register-resident recurrences' own record layout
    const size_t plain = L.BT * 4 * L.H, rec = (x3 && !L.gen_gru) ? grux_gates_floats(d->B, d->T, d->H, d->io)
                                                                   : (L.rec32 ? gru_gates_floats(d->B, d->T, d->H) : 0);
    L.st_gates = o; o += al(plain > rec ? plain : rec);
  }
  L.st_yp = o; o += al((L.BT + 1) * L.Hp);   // two planes of B*T + 1 rows
  L.st_h1 = o; o += al(L.gen_gcn ? L.BT * L.I : 0);
  {  // partial pairs (sum | max) + the tag word: one pair per workgroup of the forward recurrence (B / 16, or B for gru_small)
    const size_t nb = L.small ? (size_t)gru_small_blocks(d->B) : (size_t)grux_blocks(d->B);
    L.st_stats = o; o += al(2 * nb + 4);
  }
  // split fp16 modes with the register-resident recurrence: the gate records hold r | z | gh_n only and the BPTT kernel forms
  // n = tanh(gi_n + r gh_n) from GI, so GI (which the projection GEMM writes anyway) goes into the stash, not the workspace
  L.gi_stash = (x3 && !L.gen_gru && (d->math == WGNN_MATH_F16X3 || d->math == WGNN_MATH_F16X3G)) || L.rec32;   // (and gru.hip's)
  L.st_GI = o; o += al(L.gi_stash ? L.BT * L.Gp : 0);
  L.hq = (int)rup(L.H + 1, 16);
  L.st_hprev = o; o += al(L.g32tn ? L.BT * (size_t)L.hq : 0);     // [Hprev|1] with 16-byte aligned rows (exact fp32, large B*T)
  L.stash_floats = o;
  L.dghn = (x3 && !L.gen_gru) || (L.rec32 && L.g32tn);
  // WGNN_MATH_F16X3G, from B*T = 4096 rows: dGI / dGHn travel as ONE fp16 plane (relative rounding 2^-12, independent per
  // element); it averages out in everything they feed -- the weight gradients sum B*T rows, the conv gradients B*T*S -- and
  // the lo plane is not written, staged or multiplied (error model and measured errors)
  L.dgi1 = d->math == WGNN_MATH_F16X3G && !L.gen_gru && L.BT >= 4096;
  // ... and on the wide-GRU path (BASELINE configs[4]: H = 12288, B*T = 3072 per GPU; its BPTT cell kernel still writes both
  // planes) the same three GEMMs simply leave dGI's / dGH's lo plane unread from B*T = 3072 rows: dW_ih one pass, dW_hh and dg
  // two -- 128 of that configuration's 175 ms per step are these products
  L.gen2p = d->math == WGNN_MATH_F16X3G && L.gen_gru && L.BT >= 3072;
  // (the GCN backward rounds dg to fp16 planes anyway: measured conv gradients 5.0e-6 vs 1.7e-6 in f16x3g; the one-pass
  // fp16 mode, whose own tolerance is 5e-2, takes it at every size)
  L.dg16 = (L.dgi1 || (d->math == WGNN_MATH_F16 && !L.gen_gru)) && !L.gen_gcn;
  // the one-pass fp16 mode (tolerance class 5e-2) also takes the forward's largest intermediate, GI [B*T][3H], as one fp16
  // plane: the projection GEMM writes and the recurrence reads half the bytes (rows keep the pitch Gp, in halfs)
  L.gi16 = d->math == WGNN_MATH_F16 && !L.gen_gru;   // (grux_fwd_kernel<.., X3 = false> reads fp16 rows)
  L.hn = x3 ? grux_hn(d->H) : gru_hn(d->H);
  L.msplit = x3 ? grux_msplit(d->H) : gru_msplit(d->H);
  // GEMM rows of the dW_hh product: [dGI_r | dGI_z | pad to msplit | dGHn]
  L.m_hh = L.dghn ? (x3 ? L.msplit + L.hn : L.msplit + (int)L.H) : (int)L.G3;
  if (x3) {
    L.sk_ih = pick_splitk(L.BT, pgemm_tn_tiles((int)L.G3, (int)L.I + 1), 256, 64);   // one workgroup per CU
    L.sk_hh = pick_splitk(L.BT, pgemm_tn_tiles(L.m_hh, (int)L.H + 1), 256, 64);
  } else {
    // K chunks of at least 128 rows (B*T = 6144 at BASELINE configs[1]: with 256-row chunks the dW_hh product had 72 workgroups)
    L.sk_ih = L.g32tn ? pick_splitk(L.BT, gemm32_tn_tiles((int)L.G3, (int)L.I + 1), 256, 64)   // one workgroup per CU
                    : pick_splitk(L.BT, gemm_f32_tiles((int)L.G3, (int)L.I + 1), 1024, 128);
    L.sk_hh = L.g32tn ? pick_splitk(L.BT, gemm32_tn_tiles(L.m_hh, (int)L.H + 1), 256, 64)
                    : pick_splitk(L.BT, gemm_f32_tiles((int)L.G3, (int)L.H + 1), 1024, 128);
  }
  size_t part_ih = (size_t)L.sk_ih * L.G3 * (L.g32tn ? (size_t)gemm32_tn_pitch((int)L.I + 1) : L.I + 1);
  size_t part_hh = (size_t)L.sk_hh * (size_t)L.m_hh * (L.g32tn ? (size_t)gemm32_tn_pitch((int)L.H + 1) : L.H + 1);
  if (x3) {
    part_ih = pgemm_tn_partial_floats((int)L.G3, (int)L.I + 1, L.sk_ih);
    part_hh = pgemm_tn_partial_floats(L.m_hh, (int)L.H + 1, L.sk_hh);
  }
  o = HDR;
  L.ws_dGI = o; o += al(L.BT * L.Gp);   // fp32, or hi+lo fp16 planes (same bytes)
  L.ws_dGH = o; o += al(L.dghn ? L.BT * (size_t)L.hn : L.BT * L.Gp);   // dGHn planes, or full dGH (general GRU / f32)
  L.ws_dg = o; o += al(L.BT * L.Id);
  L.ws_part_ih = o; o += al(part_ih);     // separate regions: with WGNN_BWD_DEFER both stay live until wgnn_finish
  L.ws_part_hh = o; o += al(part_hh);
  {
    size_t a = gcn32_bwd_partial_floats((int)L.BT), b = gcnx2_bwd_partial_floats((int)L.BT);
    if (L.gen_gcn) a = gcn_csr_bwd_partial_floats();
    b *= WGNN_BWD2_MAX_CHUNKS;                                         // (chunked part 2: one set of partial rows per chunk)
    L.ws_gcnpart = o; o += al(a > b ? a : b);
  }
  L.ws_du = o; o += al(L.gen_gcn ? L.BT * L.I : 0);
  L.ws_dhz = o; o += al(L.gen_gru ? (size_t)d->B * L.H : 0);
  L.ws_dhw = o; o += al(L.gen_gru ? (size_t)d->B * L.H : 0);
  L.ws_hhp_b = o; o += al(L.gen_gru && x3 ? (size_t)L.np_h * L.Gp : 0);     // split(W_hh^T)
  L.ws_kp_b = o; o += al(L.gen_gru && x3 ? pgemm_nt_kpart_floats(d->B, (int)L.H, (int)L.Gp) : 0);
  L.ws_dc = o; o += al(L.gen_gru && x3 ? (size_t)d->B * L.Gp : 0);          // compact planes of dgh_t
  L.ws_planes_b = o; o += al(planes_b);
  L.ws_scales = o; o += al(4096);          // 3 scales, then up to 2 x 1024 block partials from offset 64
  L.ws_dY = o; o += al(L.BT * L.H);          // wgnn_bwd_mse_part outside the fused kernel: dY lives here
  L.ws_xtail_b = o; o += al((L.I & 1) && !L.gen_gcn ? L.I + 1 : 0);
  L.ws_aimg_b = o; o += al(x3 ? pgemm_nt256_aimg_bytes((int)L.BT, (int)L.I, (int)L.Gp, 2) / 4 : 0);   // dGI's planes as an image (dg GEMM)
  {
    const int sk = (!x3 && !L.g32 && L.BT < 65536) ? gemm_f32_nt_splitk((int)L.BT, (int)L.I, (int)L.G3) : 1;
    L.ws_ntk_b = o; o += al(sk > 1 ? (size_t)sk * L.BT * L.I : 0);
  }
  L.bwd_floats = o;
  return L;
}

int bwd2_chunks(const Layout& L);

// The reduction half of a finish launch: which & 4 -> the split-K partials of the two GRU weight-gradient products,
// which & 2 -> the per-workgroup partial rows of the GCN backward; gradients go to `g`.
void fill_reduce(const Layout& L, const wgnn_dims* d, const wgnn_grads* g, float* ws, int which, FinishArgs& a) {
  a.scales = ws + L.ws_scales;
  a.status = (unsigned*)ws;
  float* gs[8] = {g->conv1_weight, g->conv1_bias, g->conv2_weight, g->conv2_bias, g->w_ih, g->w_hh, g->b_ih, g->b_hh};
  const int F = d->F;
  const int64_t n[8] = {(int64_t)F * F, F, (int64_t)F * F, F, (int64_t)L.G3 * (int64_t)L.I, (int64_t)L.G3 * (int64_t)L.H,
                        (int64_t)L.G3, (int64_t)L.G3};
  for (int t = 0; t < 8; ++t) {
    a.g[t] = gs[t];
    a.n[t] = (int)n[t];
  }
  a.I = (int)L.I;
  if (which & 4) {
    FinSeg& ih = a.ih;
    ih.partial = ws + L.ws_part_ih;
    ih.splitk = L.sk_ih;
    ih.kind = L.x3 ? 2 : 1;
    ih.Mout = ih.Mgemm = (int)L.G3; ih.Nout = (int)L.I + 1; ih.ncols = (int)L.I;
    ih.scaled = L.x3;
    FinSeg& hh = a.hh;
    hh = ih;
    hh.partial = ws + L.ws_part_hh;
    hh.splitk = L.sk_hh;
    hh.Nout = (int)L.H + 1; hh.ncols = (int)L.H;
    hh.Mgemm = L.m_hh;
    ih.pitch = L.g32tn ? gemm32_tn_pitch(ih.Nout) : ih.Nout;
    hh.pitch = L.g32tn ? gemm32_tn_pitch(hh.Nout) : hh.Nout;
    if (L.dghn) { hh.msplit = L.msplit; hh.rows1 = 2 * d->H; }     // GEMM rows [dGI_r | dGI_z | pad | dGHn]
    if (L.x3) {
      pgemm_tn_geom((int)L.G3, ih.Nout, &ih.T, &ih.nNb, &ih.ntiles);
      pgemm_tn_geom(L.m_hh, hh.Nout, &hh.T, &hh.nNb, &hh.ntiles);
    }
  }
  if (which & 2) {
    a.conv_partial = ws + L.ws_gcnpart;
    const int ch = bwd2_chunks(L);      // (the option must not change between a deferred part 2 and its wgnn_finish)
    a.conv_rows = L.gen_gcn ? gcn_csr_bwd_rows()
                            : (L.x3 ? ch * gcnx_bwd_grid((int)(L.BT / ch), d->S, d->math == WGNN_MATH_F16X3 || d->math == WGNN_MATH_F16X3G)
                                    : gcn32_bwd_grid((int)L.BT, d->S));
  }
}

// Process-wide options (wgnn_set_option / wgnn_get_option).  None of them changes a result bit.
//
// WGNN_OPT_FUSED_FWD: which forwards run the fused GCN + projection kernel (gcngi.hip).
//   1 (default)  forwards WITHOUT a stash (inference: wgnn_fwd(stash = NULL), wgnn_fwd_last) -- where it measured faster
//                (B = 4096: f16x3 238 -> 215 us, f16 + bf16 I/O 149 -> 139 us; no g plane reaches HBM);
//   2            every forward it supports, training too (there the stash copy of g makes it slower: 742 -> 770 us per step
//                in f16x3, 514.5 -> 514.4 in f16: DESIGN.md, fused front end);
//   0            never.  The results are bit-identical either way (tests/test_gpu_parity.py).
// The initial value comes from the environment variable WGNN_FUSED_FWD, read ONCE (a getenv per call let the
// path change mid-process and raced with putenv from other threads); afterwards only wgnn_set_option changes it.
std::atomic<int> g_opt[WGNN_OPT_COUNT];
std::once_flag g_opt_once;

int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  if (!e || !e[0]) return dflt;
  char* end = nullptr;
  const long v = strtol(e, &end, 10);
  return (end == e || v < lo || v > hi) ? dflt : (int)v;
}

void init_options() {
  std::call_once(g_opt_once, [] {
    for (int k = 0; k < WGNN_OPT_COUNT; ++k) g_opt[k].store(0, std::memory_order_relaxed);
    g_opt[WGNN_OPT_FUSED_FWD].store(env_int("WGNN_FUSED_FWD", 1, 0, 2), std::memory_order_relaxed);
    g_opt[WGNN_OPT_BWD2_CHUNKS].store(1, std::memory_order_relaxed);
    g_opt[WGNN_OPT_BIG_GEMM].store(1, std::memory_order_relaxed);
  });
}

int opt(int key) {
  init_options();
  return g_opt[key].load(std::memory_order_relaxed);
}

// WGNN_OPT_BWD2_CHUNKS: backward part 2 (dg GEMM -> GCN backward) as C producer -> consumer pairs over C row chunks, so that a
// chunk of dg is read back while it is still cache-resident.  Only the dense fp16-plane path, only when
// every chunk still fills the chip: whole 192-row GEMM tiles, >= 12 GCN tiles per workgroup.  Else 1.
int bwd2_chunks(const Layout& L) {
  const int c = opt(WGNN_OPT_BWD2_CHUNKS);
  if (c <= 1 || !L.x3 || L.gen_gcn || L.gen_gru) return 1;
  if (L.BT % (size_t)c != 0 || (L.BT / c) % 192 != 0 || L.BT / c < 3072) return 1;
  return c;
}

int fused_fwd_mode() {
  init_options();
  return g_opt[WGNN_OPT_FUSED_FWD].load(std::memory_order_relaxed);
}

int check_dims(const wgnn_dims* d) {
  if (!d) return WGNN_ERR_NULL;
  if (d->B < 1 || d->T < 1 || d->S < 1 || d->H < 1) return WGNN_ERR_SHAPE;
  if (d->F != 13) return WGNN_ERR_SHAPE;            // the reference hard-codes 13 (step6:16)
  if ((int64_t)d->B * d->T > (1 << 30)) return WGNN_ERR_SHAPE;
  if (d->math != WGNN_MATH_F32 && d->math != WGNN_MATH_F16X3 && d->math != WGNN_MATH_F16 && d->math != WGNN_MATH_F16X3G)
    return WGNN_ERR_DTYPE;
  if (d->io != WGNN_IO_F32 && d->io != WGNN_IO_F16 && d->io != WGNN_IO_BF16) return WGNN_ERR_DTYPE;
  // 16-bit X / Y / labels: only the fp16-plane kernel family with the dense LDS-resident GCN and the register-resident GRU
  if (d->io != WGNN_IO_F32 &&
      (d->math == WGNN_MATH_F32 || d->adj_format != WGNN_ADJ_DENSE || d->S > 64 || !grux_shape_supported(d->H)))
    return WGNN_ERR_UNSUPPORTED;
  if (d->adj_format == WGNN_ADJ_CSR) {
    if (d->nnz < 1 || (int64_t)d->nnz > (int64_t)d->S * d->S) return WGNN_ERR_SHAPE;
  } else if (d->adj_format == WGNN_ADJ_DENSE) {
    if (d->S > 64) return WGNN_ERR_UNSUPPORTED;     // dense adjacency is the LDS-resident path; larger graphs: CSR
  } else {
    return WGNN_ERR_UNSUPPORTED;
  }
  // 32-bit element counts inside the kernels: one weight matrix / one activation matrix must stay below 2^31
  if ((int64_t)3 * d->H * d->S * d->F >= (1ll << 31) || (int64_t)3 * d->H * d->H >= (1ll << 31)) return WGNN_ERR_SHAPE;
  if ((int64_t)d->B * d->T * d->S * d->F >= (1ll << 31) || (int64_t)d->B * d->T * 4 * d->H >= (1ll << 31))
    return WGNN_ERR_SHAPE;
  return WGNN_OK;
}

}  // namespace

int opt_big_gemm() { return opt(WGNN_OPT_BIG_GEMM); }
int opt_gemm32_form() { return opt(WGNN_OPT_GEMM32_FORM); }

extern "C" {

int wgnn_version(void) { return WGNN_VERSION; }

int wgnn_get_option(int key) {
  if (key < 0 || key >= WGNN_OPT_COUNT) return WGNN_ERR_SHAPE;
  init_options();
  return g_opt[key].load(std::memory_order_relaxed);
}

int wgnn_set_option(int key, int value) {
  if (key < 0 || key >= WGNN_OPT_COUNT) return WGNN_ERR_SHAPE;
  if (key == WGNN_OPT_FUSED_FWD && (value < 0 || value > 2)) return WGNN_ERR_SHAPE;
  if ((key == WGNN_OPT_GG_ROLE_SPLIT && (value < 0 || value > 1)) || (key == WGNN_OPT_GG_GEMM_PRIO && (value < 0 || value > 3)))
    return WGNN_ERR_SHAPE;
  if (key == WGNN_OPT_BIG_GEMM && (value < 0 || value > 1)) return WGNN_ERR_SHAPE;
  if (key == WGNN_OPT_GEMM32_FORM && (value < 0 || value > 34)) return WGNN_ERR_SHAPE;
  if (key == WGNN_OPT_BWD2_CHUNKS && value != 1 && value != 2 && value != 4 && value != WGNN_BWD2_MAX_CHUNKS) return WGNN_ERR_SHAPE;
  init_options();
  return g_opt[key].exchange(value, std::memory_order_relaxed);
}

const char* wgnn_strerror(int status) {
  switch (status) {
    case WGNN_OK: return "ok";
    case WGNN_ERR_NULL: return "required pointer is NULL";
    case WGNN_ERR_SHAPE: return "invalid shape (need B,T,S,H >= 1 and F == 13)";
    case WGNN_ERR_DTYPE: return "unsupported dtype / math mode";
    case WGNN_ERR_WORKSPACE: return "workspace or stash too small";
    case WGNN_ERR_UNSUPPORTED:
      return "configuration not supported by this build (a dense adjacency needs S <= 64: pass larger graphs as CSR)";
    case WGNN_ERR_HIP: return "HIP runtime error (kernel launch failed)";
    case WGNN_ERR_RANGE:
      return "a value left fp16's range in an fp16-plane math mode (status block bits: 1 activation, 2 weight, 4 "
             "non-finite gradient): normalise the inputs or use WGNN_MATH_F32; or (bit 8) the backward was told the "
             "forward's loss statistics are in the stash but the last forward on it was not wgnn_fwd_loss";
    default: return "unknown status";
  }
}

size_t wgnn_workspace_bytes(const wgnn_dims* d) {
  if (check_dims(d) != WGNN_OK) return 0;
  Layout L = make_layout(d);
  return sizeof(float) * (L.fwd_floats > L.bwd_floats ? L.fwd_floats : L.bwd_floats);
}

size_t wgnn_stash_bytes(const wgnn_dims* d) {
  if (check_dims(d) != WGNN_OK) return 0;
  return sizeof(float) * make_layout(d).stash_floats;
}

// last != nullptr (wgnn_fwd_last): no stash; last[B][H] = Y[:, T-1, :] * y_mul + y_add and Y itself is only written
// where the kernels cannot skip it (into the workspace, for the exact-fp32 and general-shape recurrences).
// g_in != nullptr (wgnn_gru_fwd): the recurrent half alone on a caller-supplied g [B*T][S*F]; A, X and the conv slots of p unused.
static int fwd_impl(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p, const void* labels,
                    void* Y, void* stash, void* workspace, size_t workspace_bytes, void* stream, float* last = nullptr,
                    float wind_min = 0.f, float wind_max = 1.f, const float* g_in = nullptr) {
  // every read-out path forms its multiplier the same way, (wind_max - wind_min) in fp32, from the caller's two values
  const float y_mul = wind_max - wind_min, y_add = wind_min;
  int rc = check_dims(d);
  if (rc != WGNN_OK) return rc;
  if ((!g_in && (!A || !X)) || !p || (!Y && !last) || !workspace) return WGNN_ERR_NULL;
  if ((!g_in && (!p->conv1_weight || !p->conv1_bias || !p->conv2_weight || !p->conv2_bias)) || !p->w_ih || !p->w_hh ||
      !p->b_ih || !p->b_hh)
    return WGNN_ERR_NULL;
  if (g_in && (d->math != WGNN_MATH_F32 || d->io != WGNN_IO_F32 || d->adj_format != WGNN_ADJ_DENSE)) return WGNN_ERR_UNSUPPORTED;
  const Layout L = make_layout(d);
  if (workspace_bytes < sizeof(float) * L.fwd_floats) return WGNN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  if (last && !(L.x3 && !L.gen_gru) && !L.rec32) Y = ws + L.ws_Ylast;   // these recurrences write every row: then read the last one out
  unsigned* status = (unsigned*)workspace;           // word 0 of the status block (include/windgnn.h)
  float* sf = (float*)stash;
  float* GI = (sf && L.gi_stash) ? sf + L.st_GI : ws + L.ws_GI;
  float* g = sf ? sf + L.st_g : ws + L.ws_g;
  float* gates = sf ? sf + L.st_gates : nullptr;
  const bool x3 = L.x3;                              // fp16-plane kernels
  const bool full = d->math == WGNN_MATH_F16X3 || d->math == WGNN_MATH_F16X3G;   // three-pass split products (false: one fp16 pass)
  // the staged image of [W_ih | b_ih]: the caller's (wgnn_prepare_weights / wgnn_finish keep it current) or rebuilt here
  const bool kept = p->prepared != nullptr && L.prep_kind != 0;
  float* img_f = kept ? (float*)p->prepared + L.prep_f : ws + L.ws_planes_f;

  if (x3) {
    // W_ih as stage-major fp16 planes [np_g3][Ip] with b_ih folded into column I (g's ones column)
    if (!kept) {
      rc = launch_split_weight2(p->w_ih, (int)L.G3, (int)L.I, 0, p->b_ih, (int)L.I, img_f, L.np_g3, (int)L.Ip, status, st);
      if (rc != WGNN_OK) return rc;
    }
    // Fused front end (gcngi.hip): GCN layers + input projection in one persistent kernel, g through LDS.  The results are
    // bit-identical to the two launches below (same products, same summation order); WGNN_FUSED_FWD selects (above).
    const int fmode = fused_fwd_mode();
    if (!L.gen_gcn && !L.gen_gru && (fmode == 2 || (fmode == 1 && !sf)) && gcngi_supported(d->S, d->H, full)) {
      const int planes = sf ? ((full && !L.dgi1) ? 2 : 1) : 0;     // what the backward reads of g: hi (mask, one-pass dW_ih), + lo (strict)
      rc = launch_gcngi_fwd((int)L.BT, d->S, A, X, d->io, p->conv1_weight, p->conv1_bias, p->conv2_weight, p->conv2_bias,
                            sf ? (void*)g : nullptr, (int)L.Ip, planes, img_f, L.np_g3, GI, (int)L.Gp, (int)L.G3, full, status,
                            ws + L.ws_xtail_f, st, opt(WGNN_OPT_GG_ROLE_SPLIT), opt(WGNN_OPT_GG_GEMM_PRIO));
      if (rc != WGNN_OK) return rc;
      if (last)
        return launch_grux_fwd(d->B, d->T, d->H, GI, (int)L.Gp, p->w_hh, p->b_hh, last, nullptr, nullptr, full, status,
                               nullptr, nullptr, 0, 1, y_mul, y_add, st);
      return launch_grux_fwd(d->B, d->T, d->H, GI, (int)L.Gp, p->w_hh, p->b_hh, Y, gates, sf ? sf + L.st_yp : nullptr,
                             full, status, sf ? labels : nullptr, sf ? sf + L.st_stats : nullptr, d->io, 0, 1.f, 0.f, st);
    }
    if (L.gen_gcn)    // CSR adjacency: fp32 SpMM layers, layer 2 writes the g planes
      rc = launch_gcn2_csr_fwd((int)L.BT, d->S, d->nnz, A, (const float*)X, p->conv1_weight, p->conv1_bias,
                               p->conv2_weight, p->conv2_bias, sf ? sf + L.st_h1 : ws + L.ws_h1, nullptr, g, L.Ip, full,
                               status, st);
    else
      rc = launch_gcnx2_fwd((int)L.BT, d->S, A, X, d->io, p->conv1_weight, p->conv1_bias, p->conv2_weight,
                            p->conv2_bias, g, (int)L.Ip, full, status, ws + L.ws_xtail_f, st);
    if (rc != WGNN_OK) return rc;
    const _Float16* ghi = (const _Float16*)g;
    const size_t aimg_f = pgemm_nt256_aimg_bytes((int)L.BT, (int)L.G3, (int)L.Ip, 2);
    rc = launch_pgemm_nt(ghi, ghi + L.BT * L.Ip, (int)L.Ip, (int)L.BT, (int)L.Ip, img_f, L.np_g3, GI,
                         (int)L.Gp, (int)L.G3, nullptr, full, nullptr, st, L.gi16, aimg_f ? ws + L.ws_aimg_f : nullptr);
    if (rc != WGNN_OK) return rc;
    if (L.gen_gru) {  // any hidden width: one plane GEMM per step against split(W_hh | b_hh)
      rc = launch_split_weight2(p->w_hh, (int)L.G3, (int)L.H, 0, p->b_hh, (int)L.H, ws + L.ws_hhp_f, L.np_g3, (int)L.Hp,
                                status, st);
      if (rc != WGNN_OK) return rc;
      rc = launch_gru_gen_fwd_x3(d->B, d->T, d->H, GI, (int)L.Gp, ws + L.ws_hhp_f, L.np_g3, p->b_hh, (float*)Y, gates,
                                 sf ? sf + L.st_yp : ws + L.ws_yp, ws + L.ws_gh, ws + L.ws_kp_f, ws + L.ws_hc, full,
                                 st);
      if (rc != WGNN_OK || !last) return rc;
      return wgnn_predict_last((const float*)Y, d->B, d->T, d->H, wind_min, wind_max, last, stream);
    }
    if (last)   // the register-resident recurrence writes the read-out itself
      return launch_grux_fwd(d->B, d->T, d->H, GI, (int)L.Gp, p->w_hh, p->b_hh, last, nullptr, nullptr, full, status,
                             nullptr, nullptr, 0, 1, y_mul, y_add, st);
    // labels (wgnn_fwd_loss): the recurrence also leaves the MSE partial sums / maxima of (Y - labels) in the stash
    // (a stash without labels gets its tag word cleared: bit 8 of a later backward cannot trust stale statistics)
    return launch_grux_fwd(d->B, d->T, d->H, GI, (int)L.Gp, p->w_hh, p->b_hh, Y, gates, sf ? sf + L.st_yp : nullptr,
                           full, status, sf ? labels : nullptr, sf ? sf + L.st_stats : nullptr, d->io, 0, 1.f, 0.f, st);
  }
  if (L.gen_gcn) {
    float* h1 = sf ? sf + L.st_h1 : ws + L.ws_h1;    // layer-1 activations: kept for the backward if there is a stash
    rc = launch_gcn2_csr_fwd((int)L.BT, d->S, d->nnz, A, (const float*)X, p->conv1_weight, p->conv1_bias,
                             p->conv2_weight, p->conv2_bias, h1, g, nullptr, L.Ip, false, nullptr, st);
  } else if (g_in) {
    rc = launch_pack_g(g_in, L.BT, (int)L.I, g, (int)L.Ip, st);
  } else {
    rc = launch_gcn32_fwd((int)L.BT, d->S, A, (const float*)X, p->conv1_weight, p->conv1_bias, p->conv2_weight, p->conv2_bias, g,
                         (int)L.Ip, ws + L.ws_xtail_f, st);
  }
  if (rc != WGNN_OK) return rc;
  if (L.g32) {       // GI = [g|1] [W_ih|b_ih]^T
    float* wp = img_f;
    if (!kept) {
      rc = launch_pad_weight(p->w_ih, (int)L.G3, (int)L.I, 0, p->b_ih, wp, gemm32_nt_rows((int)L.G3), (int)L.Ip, st);
      if (rc != WGNN_OK) return rc;
    }
    rc = launch_gemm32_nt(g, (int)L.Ip, (int)L.BT, (int)L.Ip, wp, GI, (int)L.Gp, (int)L.G3, st);
  } else {
    GemmArgs ga = {};
    ga.A = g; ga.lda = (int)L.Ip; ga.a_kcontig = 1;
    ga.B = p->w_ih; ga.ldb = (int)L.I; ga.b_kcontig = 1;
    ga.C = GI; ga.ldc = (int)L.Gp; ga.M = (int)L.BT; ga.N = (int)L.G3; ga.K = (int)L.I;
    ga.bias = p->b_ih; ga.splitk = 1;
    // few rows (the reference's own call shape: 168): split-K + a fixed-order sum, so that more than 15 workgroups work
    rc = launch_gemm_f32_nt(ga, gemm_f32_nt_splitk(ga.M, ga.N, ga.K) > 1 && L.BT < 65536 ? ws + L.ws_ntk_f : nullptr, st);
  }
  if (rc != WGNN_OK) return rc;
  float* hprev = (sf && L.g32tn) ? sf + L.st_hprev : nullptr;      // [Hprev | 1 | 0..] rows for the backward's dW_hh GEMM
  if (L.gen_gru)
    rc = launch_gru_gen_fwd(d->B, d->T, d->H, GI, (int)L.Gp, p->w_hh, p->b_hh, (float*)Y, gates, ws + L.ws_gh, st);
  else if (L.small)   // few windows: one per workgroup instead of sixteen
    rc = launch_gru_small_fwd(d->B, d->T, d->H, GI, (int)L.Gp, p->w_hh, p->b_hh, (float*)Y, gates, hprev, L.hq,
                              sf ? (const float*)labels : nullptr, sf ? sf + L.st_stats : nullptr, st);
  else if (last)      // the register-resident recurrence writes the read-out itself
    return launch_gru_fwd(d->B, d->T, d->H, GI, (int)L.Gp, p->w_hh, p->b_hh, last, nullptr, nullptr, nullptr, nullptr, 0, 1,
                          y_mul, y_add, st);
  else                // labels (wgnn_fwd_loss): the recurrence also leaves the MSE partial sums of (Y - labels) in the stash
    return launch_gru_fwd(d->B, d->T, d->H, GI, (int)L.Gp, p->w_hh, p->b_hh, (float*)Y, gates,
                          sf ? (const float*)labels : nullptr, sf ? sf + L.st_stats : nullptr, hprev, L.hq, 0, 1.f, 0.f, st);
  if (rc != WGNN_OK || !last) return rc;
  return wgnn_predict_last((const float*)Y, d->B, d->T, d->H, wind_min, wind_max, last, stream);
}

int wgnn_fwd(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p, void* Y, void* stash,
             void* workspace, size_t workspace_bytes, void* stream) {
  return fwd_impl(d, A, X, p, nullptr, Y, stash, workspace, workspace_bytes, stream);
}

int wgnn_fwd_loss(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p, const void* labels,
                  void* Y, void* stash, void* workspace, size_t workspace_bytes, void* stream) {
  if (!labels || !stash) return WGNN_ERR_NULL;
  return fwd_impl(d, A, X, p, labels, Y, stash, workspace, workspace_bytes, stream);
}

int wgnn_fwd_last(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p, float wind_min,
                  float wind_max, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!out) return WGNN_ERR_NULL;
  if (d && d->io != WGNN_IO_F32) return WGNN_ERR_UNSUPPORTED;
  return fwd_impl(d, A, X, p, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream, out, wind_min, wind_max);
}

size_t wgnn_prepared_bytes(const wgnn_dims* d) {
  if (check_dims(d) != WGNN_OK) return 0;
  return sizeof(float) * make_layout(d).prep_floats;
}

int wgnn_prepare_weights(const wgnn_dims* d, const wgnn_params* p, void* workspace, size_t workspace_bytes,
                         void* stream) {
  int rc = check_dims(d);
  if (rc != WGNN_OK) return rc;
  if (!p || !p->w_ih || !p->b_ih || !p->prepared || !workspace) return WGNN_ERR_NULL;
  if (workspace_bytes < WGNN_STATUS_BYTES) return WGNN_ERR_WORKSPACE;
  const Layout L = make_layout(d);
  if (L.prep_kind == 0) return WGNN_ERR_UNSUPPORTED;            // wgnn_prepared_bytes() said 0
  hipStream_t st = (hipStream_t)stream;
  float* img_f = (float*)p->prepared + L.prep_f;
  float* img_b = (float*)p->prepared + L.prep_b;
  if (L.prep_kind == 1) {
    rc = launch_split_weight2(p->w_ih, (int)L.G3, (int)L.I, 0, p->b_ih, (int)L.I, img_f, L.np_g3, (int)L.Ip,
                              (unsigned*)workspace, st);
    if (rc != WGNN_OK) return rc;
    return launch_split_weight2(p->w_ih, (int)L.G3, (int)L.I, 1, nullptr, 0, img_b, L.np_i, (int)L.Gp, (unsigned*)workspace, st);
  }
  rc = launch_pad_weight(p->w_ih, (int)L.G3, (int)L.I, 0, p->b_ih, img_f, gemm32_nt_rows((int)L.G3), (int)L.Ip, st);
  if (rc != WGNN_OK) return rc;
  return launch_pad_weight(p->w_ih, (int)L.G3, (int)L.I, 1, nullptr, img_b, gemm32_nt_rows((int)L.I), (int)L.Gp, st);
}

int wgnn_finish(const wgnn_dims* d, const wgnn_params* p, const wgnn_grads* g, int which, const wgnn_adam* adam,
                void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_dims(d);
  if (rc != WGNN_OK) return rc;
  const int only = which & (WGNN_FINISH_ADAM_GRU | WGNN_FINISH_ADAM_CONV);   // optimiser step of one tensor family
  if ((which & ~(6 | WGNN_FINISH_ADAM_GRU | WGNN_FINISH_ADAM_CONV)) != 0 || (which == 0 && !adam)) return WGNN_ERR_SHAPE;
  if (only && (!adam || (which & 6) != 0 || only == (WGNN_FINISH_ADAM_GRU | WGNN_FINISH_ADAM_CONV))) return WGNN_ERR_SHAPE;
  if (!g || !workspace || (adam && !p)) return WGNN_ERR_NULL;
  if (!g->conv1_weight || !g->conv1_bias || !g->conv2_weight || !g->conv2_bias || !g->w_ih || !g->w_hh || !g->b_ih ||
      !g->b_hh)
    return WGNN_ERR_NULL;
  const Layout L = make_layout(d);
  if (workspace_bytes < sizeof(float) * L.bwd_floats) return WGNN_ERR_WORKSPACE;
  float* ws = (float*)workspace;
  FinishArgs a = {};
  fill_reduce(L, d, g, ws, which, a);
  if (adam) {
    if (!p->conv1_weight || !p->conv1_bias || !p->conv2_weight || !p->conv2_bias || !p->w_ih || !p->w_hh || !p->b_ih ||
        !p->b_hh)
      return WGNN_ERR_NULL;
    if (adam->step < 1) return WGNN_ERR_SHAPE;
    const float* ps[8] = {p->conv1_weight, p->conv1_bias, p->conv2_weight, p->conv2_bias, p->w_ih, p->w_hh, p->b_ih, p->b_hh};
    const wgnn_grads& m = adam->exp_avg;
    const wgnn_grads& v = adam->exp_avg_sq;
    float* ms[8] = {m.conv1_weight, m.conv1_bias, m.conv2_weight, m.conv2_bias, m.w_ih, m.w_hh, m.b_ih, m.b_hh};
    float* vs[8] = {v.conv1_weight, v.conv1_bias, v.conv2_weight, v.conv2_bias, v.w_ih, v.w_hh, v.b_ih, v.b_hh};
    for (int t = 0; t < 8; ++t) {
      if (!ms[t] || !vs[t]) return WGNN_ERR_NULL;
      a.p[t] = const_cast<float*>(ps[t]);          // the optimiser updates the parameters in place
      a.m[t] = ms[t];
      a.v[t] = vs[t];
    }
    a.adam = 1;
    const double bc1 = 1.0 - pow((double)adam->beta1, (double)adam->step);
    const double bc2 = 1.0 - pow((double)adam->beta2, (double)adam->step);
    a.lr_over_bc1 = (float)(adam->lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.b1 = adam->beta1; a.b2 = adam->beta2; a.eps = adam->eps;
    a.elem_mask = ((which & 4) ? 0u : 0xF0u) | ((which & 2) ? 0u : 0x0Fu);   // tensors whose gradient is final in g
    if (only == WGNN_FINISH_ADAM_GRU) a.elem_mask = 0xF0u;
    if (only == WGNN_FINISH_ADAM_CONV) a.elem_mask = 0x0Fu;
    if (p->prepared && L.prep_kind) {
      a.prep_kind = L.prep_kind;
      float* img_f = (float*)p->prepared + L.prep_f;
      float* img_b = (float*)p->prepared + L.prep_b;
      if (L.prep_kind == 1) {
        a.pf_hi = (_Float16*)img_f; a.pf_lo = a.pf_hi + (size_t)L.np_g3 * L.Ip;
        a.pb_hi = (_Float16*)img_b; a.pb_lo = a.pb_hi + (size_t)L.np_i * L.Gp;
        a.np_g3 = L.np_g3; a.np_i = L.np_i;
      } else {
        a.wp = img_f; a.wt = img_b;
      }
      a.Ip = (int)L.Ip; a.Gp = (int)L.Gp;
    }
  }
  return launch_finish(a, (hipStream_t)stream);
}

int wgnn_bwd(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p, const void* Y,
             const float* dY, const void* stash, const wgnn_grads* g, void* workspace, size_t workspace_bytes,
             void* stream) {
  return wgnn_bwd_part(d, A, X, p, Y, dY, stash, g, workspace, workspace_bytes, stream, 7);
}

}  // extern "C"

namespace {
// The backward behind wgnn_bwd_part (dY given) and wgnn_bwd_mse_part (labels given: dY = 2 (Y - labels) grad_scale / n
// is never written when the register-resident f16x3 recurrence runs; loss[0] = mean((Y - labels)^2)).
// dg_out != nullptr (wgnn_gru_bwd): the recurrent half alone -- the four GRU gradients and dg [B*T][S*F]; A, X, the conv slots
// of p and of g unused.
int bwd_impl(const wgnn_dims* d, const float* A, const void* Xv, const wgnn_params* p, const void* Yv,
             const float* dY, const void* labelsv, float grad_scale, float* loss, const void* stash,
             const wgnn_grads* g, void* workspace, size_t workspace_bytes, void* stream, int which, float* dg_out = nullptr) {
  if (which < 1 || which > 31 || (which & 7) == 0) return WGNN_ERR_SHAPE;
  const bool do_rec = which & 1, do_gcn = which & 2, do_wg = which & 4;
  const bool defer = which & WGNN_BWD_DEFER;          // partial sums stay in the workspace for wgnn_finish
  const float* X = (const float*)Xv;                 // io-typed (d->io): only the kernels that take `io` see 16-bit data
  const float* Y = (const float*)Yv;
  const float* labels = (const float*)labelsv;
  const bool stats_ready = (which & 8) && labels;   // wgnn_fwd_loss left the MSE partials in the stash
  int rc = check_dims(d);
  if (rc != WGNN_OK) return rc;
  if ((!dg_out && (!A || !X)) || !p || !Y || (!dY && !labels) || !stash || !g || !workspace) return WGNN_ERR_NULL;
  if (labels && do_rec && !loss) return WGNN_ERR_NULL;
  if ((!dg_out && (!g->conv1_weight || !g->conv1_bias || !g->conv2_weight || !g->conv2_bias)) || !g->w_ih || !g->w_hh ||
      !g->b_ih || !g->b_hh)
    return WGNN_ERR_NULL;
  if (dg_out && (d->math != WGNN_MATH_F32 || d->io != WGNN_IO_F32 || d->adj_format != WGNN_ADJ_DENSE || which != 7))
    return WGNN_ERR_UNSUPPORTED;
  const Layout L = make_layout(d);
  if (workspace_bytes < sizeof(float) * L.bwd_floats) return WGNN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  unsigned* status = (unsigned*)workspace;
  const float* sf = (const float*)stash;
  const float* gact = sf + L.st_g;
  const float* gates = sf + L.st_gates;
  float* dGI = ws + L.ws_dGI;
  float* dGH = ws + L.ws_dGH;
  float* dg = ws + L.ws_dg;
  float* part_ih = ws + L.ws_part_ih;
  float* part_hh = ws + L.ws_part_hh;
  const bool kept = p->prepared != nullptr && L.prep_kind != 0;
  float* img_b = kept ? (float*)p->prepared + L.prep_b : ws + L.ws_planes_b;     // staged image of W_ih^T
  float* scales = ws + L.ws_scales;          // [0] = 2^k, [1] = 2^-k (f16x3 range scaling), [2] = dY coefficient; partials from 64
  const bool x3 = L.x3;
  const bool full = d->math == WGNN_MATH_F16X3 || d->math == WGNN_MATH_F16X3G;
  auto reduce_now = [&](int parts) {                 // without WGNN_BWD_DEFER: the reduce-only form of the finish launch
    FinishArgs fa = {};
    fill_reduce(L, d, g, ws, parts, fa);
    return launch_finish(fa, st);
  };
  // the recurrence kernel forms dY from the labels itself: the fast f16x3 recurrence always (its statistics pass is
  // cheap), the exact-fp32 register-resident one when the forward left the loss statistics in the stash
  const bool fused_loss = labels && ((x3 && !L.gen_gru) || ((L.rec32 || L.small) && stats_ready));
  // 16-bit labels / Y: the statistics must come from wgnn_fwd_loss (the stand-alone pass reads fp32 only)
  if (d->io != WGNN_IO_F32 && labels && do_rec && !(fused_loss && stats_ready)) return WGNN_ERR_UNSUPPORTED;
  if (labels && !fused_loss && do_rec) {                   // other kernels: materialise dY in the workspace
    rc = launch_mse(Y, labels, (int64_t)L.BT * L.H, grad_scale, ws + L.ws_dY, loss, scales + 64, st);
    if (rc != WGNN_OK) return rc;
  }
  if (labels && !fused_loss) dY = ws + L.ws_dY;

  if (x3) {
    // Everything downstream of dY is linear in it: run it in units scaled by scales[0] = 2^k (so that
    // fp16 never sees ~1e-9 values) and multiply only the final gradients by scales[1] = 2^-k.
    _Float16* dGIh = (_Float16*)dGI;
    _Float16* dGHh = (_Float16*)dGH;
    const _Float16* gh = (const _Float16*)gact;
    const _Float16* yph = (const _Float16*)(sf + L.st_yp);
    const size_t PG = L.BT * L.Gp;
    const _Float16* dGIlo = L.dgi1 ? nullptr : dGIh + PG;                    // nullptr: single-plane A operand, two passes
    const _Float16* dGHnlo = L.dgi1 ? nullptr : dGHh + L.BT * (size_t)L.hn;
    if (do_rec) {
      if (fused_loss && stats_ready)   // the forward recurrence already reduced (Y - labels): the BPTT kernel finalises
        rc = WGNN_OK;
      else if (fused_loss)   // loss, the range scale and the dY coefficient in one pass over Y and the labels
        rc = launch_mse_stats(Y, labels, (int64_t)L.BT * L.H, grad_scale, loss, scales, scales + 64, st);
      else
        rc = launch_amax_scale(dY, (int64_t)L.BT * L.H, scales, scales + 64, st);   // 448 partials after the scales
      if (rc != WGNN_OK) return rc;
      if (L.gen_gru) {
        rc = launch_split_weight2(p->w_hh, (int)L.G3, (int)L.H, 1, nullptr, 0, ws + L.ws_hhp_b, L.np_h, (int)L.Gp, status,
                                  st);
        if (rc != WGNN_OK) return rc;
        rc = launch_gru_gen_bwd_x3(d->B, d->T, d->H, ws + L.ws_hhp_b, L.np_h, Y, dY, gates, scales, dGIh, dGHh,
                                   (int)L.Gp, ws + L.ws_dhz, ws + L.ws_dhw, ws + L.ws_kp_b, ws + L.ws_dc, full, st);
      } else {
        rc = launch_grux_bwd(d->B, d->T, d->H, p->w_hh, Yv, fused_loss ? nullptr : dY, fused_loss ? labelsv : nullptr,
                             d->io, gates, L.gi_stash ? sf + L.st_GI : nullptr, (int)L.Gp, scales, dGIh, dGHh, (int)L.Gp, full,
                             fused_loss && stats_ready ? sf + L.st_stats : nullptr, (int64_t)L.BT * L.H, grad_scale, loss,
                             scales, status, L.dgi1 ? 0 : 1, st);
      }
      if (rc != WGNN_OK) return rc;
    }
    if (do_wg) {
      // dW_hh | db_hh = dGH^T [Hprev | 1]   (Hprev row (b,t) = Y-plane row (b,t-1); row B*T stands in at t = 0).
      // Register-resident recurrence: dGH = [dGI_r | dGI_z | dGHn], the A operand takes GEMM rows < msplit from the dGI
      // planes and rows >= msplit from the dGHn planes; the reduce kernel maps the GEMM rows back to W_hh's rows.
      if (L.dghn) {
        rc = launch_pgemm_tn(dGIh, dGIlo, (int)L.Gp, yph, yph + (L.BT + 1) * L.Hp, (int)L.Hp, d->T, (int)L.BT,
                             L.sk_hh, part_hh, L.m_hh, (int)L.H + 1, full, dGHh, dGHnlo, L.hn, L.msplit, st, /*b_stream=*/true);
      } else {
        rc = launch_pgemm_tn(dGHh, L.gen2p ? nullptr : dGHh + PG, (int)L.Gp, yph, yph + (L.BT + 1) * L.Hp, (int)L.Hp, d->T,
                             (int)L.BT, L.sk_hh, part_hh, (int)L.G3, (int)L.H + 1, full, nullptr, nullptr, 0, 0, st);
      }
      if (rc != WGNN_OK) return rc;
      // dW_ih | db_ih = dGI^T [g | 1]
      // (single-plane dGI: ONE pass, hi(dGI) x hi(g) -- the rounding of g, too, is independent per element and averages
      // out over the B*T rows this product sums: measured 8.8e-6 of max at B*T = 6144 against 5.9e-6 with g's lo plane.
      // The same was measured for dW_hh (7e-5: h rows are correlated) and dg (2.8e-4 on the conv gradients: the rounding
      // of W_ih is the same for every row and does not average) and NOT adopted: they keep hi x (hi + lo).)
      const bool one_pass_ih = L.dgi1 || L.gen2p;
      rc = launch_pgemm_tn(dGIh, one_pass_ih ? dGIh : dGIh + PG, (int)L.Gp, gh, gh + L.BT * L.Ip, (int)L.Ip, 0, (int)L.BT,
                           L.sk_ih, part_ih, (int)L.G3, (int)L.I + 1, full && !one_pass_ih, nullptr, nullptr, 0, 0, st,
                           /*b_stream=*/true);
      if (rc != WGNN_OK) return rc;
      if (!defer) rc = reduce_now(4);
      if (rc != WGNN_OK) return rc;
    }   // the four GRU gradients are final here: a data-parallel caller can start reducing them now
    if (!do_gcn) return WGNN_OK;
    // dg = dGI W_ih   (B operand = split(W_ih^T) [np_i][Gp]); dg stays in scaled units
    if (!kept) {
      rc = launch_split_weight2(p->w_ih, (int)L.G3, (int)L.I, 1, nullptr, 0, img_b, L.np_i, (int)L.Gp, status, st);
      if (rc != WGNN_OK) return rc;
    }
    const int chunks = bwd2_chunks(L);
    if (chunks > 1) {     // producer -> consumer pairs over row chunks (WGNN_OPT_BWD2_CHUNKS); same products, same partial sums per tile
      const size_t rows = L.BT / chunks;
      const size_t es = d->io ? 2 : 4, dgs = L.dg16 ? 2 : 4;
      const int prow = gcnx_bwd_grid((int)rows, d->S, full);
      for (int c = 0; c < chunks; ++c) {
        const size_t r0 = (size_t)c * rows;
        rc = launch_pgemm_nt(dGIh + r0 * L.Gp, dGIlo ? dGIlo + r0 * L.Gp : nullptr, (int)L.Gp, (int)rows, (int)L.Gp, img_b, L.np_i,
                             (float*)((char*)dg + r0 * L.Id * dgs), (int)L.Id, (int)L.I, nullptr, full, nullptr, st, L.dg16);
        if (rc != WGNN_OK) return rc;
        rc = launch_gcnx2_bwd((int)rows, d->S, A, (const char*)Xv + r0 * L.I * es, d->io, p->conv1_weight, p->conv1_bias,
                              p->conv2_weight, gh + r0 * L.Ip, (int)L.Ip, (const char*)dg + r0 * L.Id * dgs, (int)L.Id, L.dg16,
                              scales, /*scale_in=*/0, ws + L.ws_gcnpart + (size_t)c * prow * gcnx2_bwd_partial_floats((int)rows) / 256,
                              full, ws + L.ws_xtail_b, st);
        if (rc != WGNN_OK) return rc;
      }
      if (defer) return WGNN_OK;
      return reduce_now(2);
    }
    const size_t aimg_b = pgemm_nt256_aimg_bytes((int)L.BT, (int)L.I, (int)L.Gp, 2);
    rc = launch_pgemm_nt(dGIh, L.gen_gru ? (L.gen2p ? nullptr : dGIh + PG) : dGIlo, (int)L.Gp, (int)L.BT, (int)L.Gp, img_b, L.np_i, dg, (int)L.Id,
                         (int)L.I, nullptr, full, nullptr, st, L.dg16, aimg_b ? ws + L.ws_aimg_b : nullptr);
    if (rc != WGNN_OK) return rc;
    if (L.gen_gcn) {
      rc = launch_gcn2_csr_bwd((int)L.BT, d->S, d->nnz, A, X, p->conv2_weight, sf + L.st_h1, nullptr, gact, L.Ip, dg,
                                 L.Id, scales, ws + L.ws_du, ws + L.ws_gcnpart, nullptr, nullptr, nullptr, nullptr, st);
      if (rc != WGNN_OK || defer) return rc;
      return reduce_now(2);
    }
    rc = launch_gcnx2_bwd((int)L.BT, d->S, A, Xv, d->io, p->conv1_weight, p->conv1_bias, p->conv2_weight, gact, (int)L.Ip, dg, (int)L.Id, L.dg16,
                          scales, /*scale_in=*/0, ws + L.ws_gcnpart, full, ws + L.ws_xtail_b, st);
    if (rc != WGNN_OK || defer) return rc;
    return reduce_now(2);
  }

  if (do_rec) {
    if (L.gen_gru)
      rc = launch_gru_gen_bwd(d->B, d->T, d->H, p->w_hh, Y, dY, gates, dGI, dGH, (int)L.Gp, ws + L.ws_dhz,
                              ws + L.ws_dhw, st);
    else if (L.small)
      rc = launch_gru_small_bwd(d->B, d->T, d->H, p->w_hh, Y, fused_loss ? nullptr : dY, fused_loss ? labels : nullptr, gates,
                                dGI, dGH, (int)L.Gp, fused_loss ? sf + L.st_stats : nullptr, (int64_t)L.BT * L.H, grad_scale,
                                loss, status, st);
    else   // register-resident recurrence: dGHn alone when the dW_hh GEMM has the two-source A operand; fused loss
      rc = launch_gru_bwd(d->B, d->T, d->H, p->w_hh, Y, fused_loss ? nullptr : dY, fused_loss ? labels : nullptr, gates,
                          sf + L.st_GI, (int)L.Gp, dGI, (int)L.Gp, L.dghn ? dGH : nullptr, L.dghn ? nullptr : dGH,
                          fused_loss ? sf + L.st_stats : nullptr, (int64_t)L.BT * L.H, grad_scale, loss, status, st);
    if (rc != WGNN_OK) return rc;
  }
  if (do_wg) {
    // dW_hh = dGH^T Hprev, db_hh = dGH^T 1   (Hprev row (b,t) = Y row (b,t-1), zero at t = 0)
    if (L.g32tn) {     // [Hprev | 1 | 0..] rows were written by the forward recurrence (stash)
      const float* hp = sf + L.st_hprev;
      if (L.dghn)      // dGH = [dGI_r | dGI_z | dGHn]: GEMM rows < msplit from dGI, the rest from dGHn
        rc = launch_gemm32_tn(dGI, (int)L.Gp, hp, L.hq, (int)L.BT, L.sk_hh, part_hh, L.m_hh, (int)L.H + 1, dGH, L.hn,
                              L.msplit, st);
      else
        rc = launch_gemm32_tn(dGH, (int)L.Gp, hp, L.hq, (int)L.BT, L.sk_hh, part_hh, (int)L.G3, (int)L.H + 1, nullptr, 0, 0,
                              st);
    } else {
      GemmArgs a = {};
      a.A = dGH; a.lda = (int)L.Gp; a.a_kcontig = 0;
      a.B = Y; a.ldb = (int)L.H; a.b_kcontig = 0; a.ones_col = 1; a.shift_T = d->T;
      a.M = (int)L.G3; a.N = (int)L.H + 1; a.K = (int)L.BT;
      a.splitk = L.sk_hh; a.partial = part_hh;
      rc = launch_gemm_f32(a, st);
    }
    if (rc != WGNN_OK) return rc;
    // dW_ih = dGI^T g, db_ih = dGI^T 1
    if (L.g32tn) {     // g carries its ones column (gcn32_fwd)
      rc = launch_gemm32_tn(dGI, (int)L.Gp, gact, (int)L.Ip, (int)L.BT, L.sk_ih, part_ih, (int)L.G3, (int)L.I + 1, nullptr, 0,
                            0, st);
    } else {
      GemmArgs b = {};
      b.A = dGI; b.lda = (int)L.Gp; b.a_kcontig = 0;
      b.B = gact; b.ldb = (int)L.Ip; b.b_kcontig = 0; b.ones_col = 1;
      b.M = (int)L.G3; b.N = (int)L.I + 1; b.K = (int)L.BT;
      b.splitk = L.sk_ih; b.partial = part_ih;
      rc = launch_gemm_f32(b, st);
    }
    if (rc != WGNN_OK) return rc;
    if (!defer) rc = reduce_now(4);
    if (rc != WGNN_OK) return rc;
  }
  if (!do_gcn) return WGNN_OK;
  {
    // dg = dGI W_ih
    if (L.g32) {
      float* wt = img_b;                  // (W_ih^T) [I -> padded][Gp]
      if (!kept) {
        rc = launch_pad_weight(p->w_ih, (int)L.G3, (int)L.I, 1, nullptr, wt, gemm32_nt_rows((int)L.I), (int)L.Gp, st);
        if (rc != WGNN_OK) return rc;
      }
      rc = launch_gemm32_nt(dGI, (int)L.Gp, (int)L.BT, (int)L.Gp, wt, dg, (int)L.Id, (int)L.I, st);
    } else {
      GemmArgs c = {};
      c.A = dGI; c.lda = (int)L.Gp; c.a_kcontig = 1;
      c.B = p->w_ih; c.ldb = (int)L.I; c.b_kcontig = 0;
      c.C = dg; c.ldc = (int)L.Id; c.M = (int)L.BT; c.N = (int)L.I; c.K = (int)L.G3; c.splitk = 1;
      rc = launch_gemm_f32_nt(c, gemm_f32_nt_splitk(c.M, c.N, c.K) > 1 && L.BT < 65536 ? ws + L.ws_ntk_b : nullptr, st);
    }
    if (rc != WGNN_OK) return rc;
  }
  if (dg_out) return launch_unpack_dg(dg, L.BT, (int)L.I, (int)L.Id, dg_out, st);     // wgnn_gru_bwd: hand dg to the caller
  if (L.gen_gcn)
    rc = launch_gcn2_csr_bwd((int)L.BT, d->S, d->nnz, A, X, p->conv2_weight, sf + L.st_h1, gact, nullptr, L.Ip, dg,
                             L.Id, nullptr, ws + L.ws_du, ws + L.ws_gcnpart, nullptr, nullptr, nullptr, nullptr, st);
  else
    rc = launch_gcn32_bwd((int)L.BT, d->S, A, X, p->conv1_weight, p->conv1_bias, p->conv2_weight, gact, (int)L.Ip, dg, (int)L.Id,
                          nullptr, nullptr, nullptr, nullptr, ws + L.ws_gcnpart, ws + L.ws_xtail_b, st);
  if (rc != WGNN_OK || defer) return rc;
  return reduce_now(2);
}
}  // namespace

extern "C" {

int wgnn_bwd_part(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p, const void* Y,
                  const float* dY, const void* stash, const wgnn_grads* g, void* workspace, size_t workspace_bytes,
                  void* stream, int which) {
  if (!dY) return WGNN_ERR_NULL;
  return bwd_impl(d, A, X, p, Y, dY, nullptr, 1.f, nullptr, stash, g, workspace, workspace_bytes, stream, which);
}

int wgnn_bwd_mse_part(const wgnn_dims* d, const float* A, const void* X, const wgnn_params* p, const void* Y,
                      const void* labels, float grad_scale, float* loss, const void* stash, const wgnn_grads* g,
                      void* workspace, size_t workspace_bytes, void* stream, int which) {
  if (!labels) return WGNN_ERR_NULL;
  return bwd_impl(d, A, X, p, Y, nullptr, labels, grad_scale, loss, stash, g, workspace, workspace_bytes, stream,
                  which);
}

int wgnn_gru_fwd(const wgnn_dims* d, const float* g, const wgnn_params* p, void* Y, void* stash, void* workspace,
                 size_t workspace_bytes, void* stream) {
  if (!g) return WGNN_ERR_NULL;
  return fwd_impl(d, nullptr, nullptr, p, nullptr, Y, stash, workspace, workspace_bytes, stream, nullptr, 0.f, 1.f, g);
}

int wgnn_gru_bwd(const wgnn_dims* d, const float* g, const wgnn_params* p, const void* Y, const float* dY, const void* stash,
                 const wgnn_grads* grads, float* dg, void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !dY || !dg) return WGNN_ERR_NULL;
  return bwd_impl(d, nullptr, nullptr, p, Y, dY, nullptr, 1.f, nullptr, stash, grads, workspace, workspace_bytes, stream, 7, dg);
}

// F == F_out == 13 (the reference's own layers, src/main.py:41): the MFMA kernels of gcn.hip; any other widths: gcn_any.hip
size_t wgnn_gcn_layer_workspace_bytes(int32_t ntiles, int32_t S, int32_t F, int32_t F_out) {
  if (ntiles < 1 || !gcn_any_supported(S, F, F_out)) return 0;
  if (F == 13 && F_out == 13) return sizeof(float) * gcn1_bwd_partial_floats(ntiles);
  return sizeof(float) * gcn_any_bwd_partial_floats(ntiles, F, F_out);
}

int wgnn_gcn_layer_fwd(int32_t ntiles, int32_t S, int32_t F, int32_t F_out, const float* A, const float* X, const float* W,
                       const float* b, float* out, void* stream) {
  if (ntiles < 1 || S < 1 || F < 1 || F_out < 1) return WGNN_ERR_SHAPE;
  if (!gcn_any_supported(S, F, F_out)) return WGNN_ERR_UNSUPPORTED;
  if (!A || !X || !W || !b || !out) return WGNN_ERR_NULL;
  if (F == 13 && F_out == 13) return launch_gcn1_fwd(ntiles, S, A, X, W, b, out, (hipStream_t)stream);
  return launch_gcn_any_fwd(ntiles, S, F, F_out, A, X, W, b, out, (hipStream_t)stream);
}

int wgnn_gcn_layer_bwd(int32_t ntiles, int32_t S, int32_t F, int32_t F_out, const float* A, const float* X, const float* W,
                       const float* out, const float* dout, float* dW, float* db, float* dX, void* workspace,
                       size_t workspace_bytes, void* stream) {
  if (ntiles < 1 || S < 1 || F < 1 || F_out < 1) return WGNN_ERR_SHAPE;
  if (!gcn_any_supported(S, F, F_out)) return WGNN_ERR_UNSUPPORTED;
  if (!A || !X || !W || !out || !dout || !dW || !db || !workspace) return WGNN_ERR_NULL;
  if (workspace_bytes < wgnn_gcn_layer_workspace_bytes(ntiles, S, F, F_out)) return WGNN_ERR_WORKSPACE;
  if (F == 13 && F_out == 13)
    return launch_gcn1_bwd(ntiles, S, A, X, W, out, dout, dW, db, dX, (float*)workspace, (hipStream_t)stream);
  return launch_gcn_any_bwd(ntiles, S, F, F_out, A, X, W, out, dout, dW, db, dX, (float*)workspace, (hipStream_t)stream);
}

size_t wgnn_gcn_layer_csr_workspace_bytes(int32_t ntiles, int32_t S, int32_t F) {
  if (ntiles < 1 || S < 1 || F != 13 || (int64_t)ntiles * S * F >= (1ll << 31)) return 0;
  return sizeof(float) * gcn1_csr_bwd_ws_floats(ntiles, S);
}

int wgnn_gcn_layer_csr_fwd(int32_t ntiles, int32_t S, int32_t F, int32_t nnz, const void* csr, const float* X,
                           const float* W, const float* b, float* out, void* stream) {
  if (ntiles < 1 || S < 1 || F != 13 || nnz < 1 || (int64_t)ntiles * S * F >= (1ll << 31)) return WGNN_ERR_SHAPE;
  if (!csr || !X || !W || !b || !out) return WGNN_ERR_NULL;
  return launch_gcn1_csr_fwd(ntiles, S, nnz, csr, X, W, b, out, (hipStream_t)stream);
}

int wgnn_gcn_layer_csr_bwd(int32_t ntiles, int32_t S, int32_t F, int32_t nnz, const void* csr, const float* X,
                           const float* W, const float* out, const float* dout, float* dW, float* db, float* dX,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (ntiles < 1 || S < 1 || F != 13 || nnz < 1 || (int64_t)ntiles * S * F >= (1ll << 31)) return WGNN_ERR_SHAPE;
  if (!csr || !X || !W || !out || !dout || !dW || !db || !workspace) return WGNN_ERR_NULL;
  if (workspace_bytes < wgnn_gcn_layer_csr_workspace_bytes(ntiles, S, F)) return WGNN_ERR_WORKSPACE;
  return launch_gcn1_csr_bwd(ntiles, S, nnz, csr, X, W, out, dout, dW, db, dX, (float*)workspace, (hipStream_t)stream);
}

int wgnn_mse_loss_grad(const float* Y, const float* L, int64_t n, float grad_scale, float* dY, float* loss,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!Y || !L || !loss || !workspace) return WGNN_ERR_NULL;
  if (n < 1) return WGNN_ERR_SHAPE;
  if (workspace_bytes < 4096) return WGNN_ERR_WORKSPACE;
  return launch_mse(Y, L, n, grad_scale, dY, loss, (float*)workspace, (hipStream_t)stream);
}

int wgnn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                   float lr, float beta1, float beta2, float eps, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return WGNN_ERR_NULL;
  if (n < 1 || step < 1) return WGNN_ERR_SHAPE;
  return launch_adam(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, (hipStream_t)stream);
}

}  // extern "C"
