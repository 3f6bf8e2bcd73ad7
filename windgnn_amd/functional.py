"""torch.autograd bindings over the C ABI.  PyTorch is plumbing here: device memory, streams and
the autograd graph; all arithmetic of the path runs in libwindgnn_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib

PARAM_ORDER = ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias",
               "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0")


_IO_OF = {torch.float32: _lib.IO_F32, torch.float16: _lib.IO_F16, torch.bfloat16: _lib.IO_BF16}


def _require_gpu(*tensors: torch.Tensor, io_ok: bool = False) -> None:
    """io_ok: the tensor is one of the "on the wire" tensors (X, Y, labels), which may also be fp16 / bf16 (wgnn_io)."""
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "windgnn_amd runs on an MI355X (HIP) only: got a %s tensor. There is no CPU fallback; "
                "use the oracle under oracle/ for CPU checks." % t.device)
        if t.dtype != torch.float32 and not (io_ok and t.dtype in _IO_OF):
            raise RuntimeError("windgnn_amd: expected float32 tensors, got %s" % t.dtype)


def _require_contiguous(**named: torch.Tensor) -> None:
    """The C ABI reads dense row-major memory from data_ptr(): a strided view (e.g. a batch slice X[::2], a transposed
    label tensor) would be read as if it were dense.  The raw entry points refuse it; callers own the .contiguous()."""
    for name, t in named.items():
        if t is not None and not t.is_contiguous():
            raise RuntimeError("windgnn_amd: %s must be contiguous (got shape %s with strides %s): call .contiguous() "
                               "on it first" % (name, tuple(t.shape), tuple(t.stride())))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _scratch(device, nbytes: int):
    """(pointer, bytes) of plain scratch for the entry points that have no status block (wgnn_mse_loss_grad,
    wgnn_gcn_layer_*_bwd): the shared workspace past its first 256 bytes."""
    ws = _Workspace.get(device, nbytes + _lib.STATUS_BYTES)
    return C.c_void_p(ws.data_ptr() + _lib.STATUS_BYTES), ws.numel() - _lib.STATUS_BYTES


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _adj(A, S=None):
    """(device tensor to pass, adj_format, nnz) for a dense [S,S] tensor or a graph.CsrAdjacency.  `S` = the station
    count of the features it will multiply: the kernels locate rowptr / col / val inside the CSR buffer from S and
    nnz alone and the C ABI cannot look into device memory, so a buffer built for another graph is refused HERE."""
    if hasattr(A, "blob"):                                    # CsrAdjacency
        if A.blob.dtype != torch.int32 or A.blob.dim() != 1 or not A.blob.is_contiguous():
            raise RuntimeError("windgnn_amd: the CSR adjacency buffer must be a contiguous 1-D int32 tensor, got %s %s"
                               % (A.blob.dtype, tuple(A.blob.shape)))
        if A.blob.numel() != 2 * (A.S + 1) + 4 * A.nnz:
            raise RuntimeError("windgnn_amd: CSR adjacency buffer has %d words, expected 2*(S+1) + 4*nnz = %d "
                               "(S=%d, nnz=%d)" % (A.blob.numel(), 2 * (A.S + 1) + 4 * A.nnz, A.S, A.nnz))
        if S is not None and A.S != S:
            raise RuntimeError("windgnn_amd: CSR adjacency of %d stations does not match %d stations" % (A.S, S))
        if not A.blob.is_cuda:
            raise RuntimeError("windgnn_amd: the CSR adjacency is on %s; call .to(device) first (no CPU fallback)"
                               % A.blob.device)
        return A.blob, _lib.ADJ_CSR, A.nnz
    _require_gpu(A)
    if S is not None and tuple(A.shape) != (S, S):
        raise RuntimeError("windgnn_amd: adjacency %s does not match %d stations" % (tuple(A.shape), S))
    return A.contiguous(), _lib.ADJ_DENSE, 0


def _params_struct(cls, tensors: Sequence[torch.Tensor], prepared: torch.Tensor = None):
    s = cls()
    for (name, _), t in zip(cls._fields_, tensors):
        setattr(s, name, t.data_ptr())
    if prepared is not None:
        s.prepared = prepared.data_ptr()      # wgnn_params.prepared: caller-kept images of W_ih
    return s


class _Workspace:
    """Per-device grow-only scratch, so the steady state does no allocation.  Allocated zeroed: its first 256 bytes
    are the library's sticky status block (include/windgnn.h), which kernels only OR into."""
    _bufs = {}

    @classmethod
    def get(cls, device, nbytes: int) -> torch.Tensor:
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        buf = cls._bufs.get(key)
        nbytes = max(nbytes, _lib.STATUS_BYTES)
        if buf is None or buf.numel() < nbytes:
            status = buf[:_lib.STATUS_BYTES].clone() if buf is not None else None
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
            if status is not None:
                buf[:_lib.STATUS_BYTES] = status            # a pending report survives the re-allocation
            cls._bufs[key] = buf
        return buf


_STATUS_TEXT = {1: "a graph-convolution pre-activation left fp16's range (|x| > 65504) or was NaN",
                2: "a GRU weight or bias lies outside fp16's range",
                4: "a gradient came out inf / NaN",
                8: "wgnn_bwd_mse_part(part | 8) ran on a stash whose last forward was not wgnn_fwd_loss: the loss is NaN "
                   "and the gradients are not to be used"}


def check_range_status(device=None) -> None:
    """Read the status word of this process's workspaces on `device` (one 4-byte device-to-host copy each, so a
    synchronisation) and raise if a kernel of the fp16-plane math modes (f16x3 / f16) reported a value it cannot
    represent.  The reference's fp32 path has no such limit; the answer is never silently inf/NaN/0."""
    for (index, _), buf in list(_Workspace._bufs.items()):
        if device is not None and torch.device(device).index not in (None, index):
            continue
        word = int(buf[:4].view(torch.int32).item())
        if word:
            buf[:4].zero_()
            what = "; ".join(t for b, t in _STATUS_TEXT.items() if word & b)
            raise RuntimeError("windgnn_amd: %s (status %d: %s). The f16x3 / f16 math modes hold activations and "
                               "weights as fp16 planes: normalise the inputs (the reference min-max-normalises "
                               "every feature to [0, 1]) or use math='f32'."
                               % (_lib.load().wgnn_strerror(-7).decode(), word, what))


def gcn_gru_forward_raw(A, X, params: Sequence[torch.Tensor], math=_lib.MATH_F32, want_stash=True, labels=None,
                        prepared=None):
    """Y[B,T,H], stash = wgnn_fwd(...).  X is [B,T,S,F].  labels [B,T,H]: wgnn_fwd_loss (the MSE statistics of
    (Y - labels) are left in the stash for gcn_gru_backward_mse_raw(..., part | 8))."""
    lib = _lib.load()
    _require_gpu(X, io_ok=True)
    _require_gpu(*params)
    _require_contiguous(X=X, labels=labels, **{"params[%d]" % i: q for i, q in enumerate(params)})
    B, T, S, F = X.shape
    A, fmt, nnz = _adj(A, S)
    H = params[5].shape[1]
    d = _lib.Dims(B, T, S, F, H, math, fmt, nnz, _IO_OF[X.dtype])     # X's dtype is the I/O type: Y comes back in it
    ws_bytes = lib.wgnn_workspace_bytes(C.byref(d))
    if ws_bytes == 0:
        _lib.check(-5 if F == 13 else -2, "wgnn_workspace_bytes(B=%d,T=%d,S=%d,F=%d,H=%d,math=%d,io=%s)"
                   % (B, T, S, F, H, math, X.dtype))
    ws = _Workspace.get(X.device, ws_bytes)
    stash = torch.empty(lib.wgnn_stash_bytes(C.byref(d)), dtype=torch.uint8, device=X.device) if want_stash else None
    Y = torch.empty(B, T, H, dtype=X.dtype, device=X.device)
    ps = _params_struct(_lib.Params, params, prepared)
    if labels is not None:
        _require_gpu(labels, io_ok=True)
        if labels.dtype != X.dtype:
            raise RuntimeError("windgnn_amd: labels are %s but X is %s (one I/O type per call)" % (labels.dtype, X.dtype))
        if labels.numel() != Y.numel() or not want_stash:
            raise RuntimeError("windgnn_amd: wgnn_fwd_loss needs a stash and labels of Y's size, got %s vs %s"
                               % (tuple(labels.shape), tuple(Y.shape)))
        rc = lib.wgnn_fwd_loss(C.byref(d), _ptr(A), _ptr(X), C.byref(ps), _ptr(labels), _ptr(Y),
                               _ptr(stash), _ptr(ws), ws_bytes, _stream())
        _lib.check(rc, "wgnn_fwd_loss")
        return Y, stash, d
    rc = lib.wgnn_fwd(C.byref(d), _ptr(A), _ptr(X), C.byref(ps), _ptr(Y), _ptr(stash), _ptr(ws), ws_bytes, _stream())
    _lib.check(rc, "wgnn_fwd")
    return Y, stash, d


def gcn_gru_backward_raw(d, A, X, params, Y, dY, stash, grads: Sequence[torch.Tensor], part: int = 7, stream=None):
    """part bit mask (wgnn_bwd_part): 1 = BPTT recurrence, 4 = GRU weight-gradient GEMMs, 2 = dg + GCN backward."""
    lib = _lib.load()
    _require_contiguous(X=X, Y=Y, dY=dY, **{"grads[%d]" % i: q for i, q in enumerate(grads)},
                        **{"params[%d]" % i: q for i, q in enumerate(params)})
    A = getattr(A, "blob", A)              # CsrAdjacency -> its device buffer (d.adj_format says which it is)
    _require_contiguous(adj_matrix=A)
    ws_bytes = lib.wgnn_workspace_bytes(C.byref(d))
    ws = _Workspace.get(X.device, ws_bytes)
    ps = _params_struct(_lib.Params, params)
    gs = _params_struct(_lib.Grads, grads)
    rc = lib.wgnn_bwd_part(C.byref(d), _ptr(A), _ptr(X), C.byref(ps), _ptr(Y), _ptr(dY), _ptr(stash), C.byref(gs),
                           _ptr(ws), ws_bytes, _stream() if stream is None else C.c_void_p(stream.cuda_stream), part)
    _lib.check(rc, "wgnn_bwd_part(%d)" % part)


def gcn_gru_backward_mse_raw(d, A, X, params, Y, L, stash, grads: Sequence[torch.Tensor], loss: torch.Tensor,
                             grad_scale: float = 1.0, part: int = 7, prepared=None):
    """wgnn_bwd_mse_part: the backward of grad_scale * mean((Y - L)^2) with the loss call folded in (src/main.py:72,79);
    `loss` (0-dim device tensor) receives mean((Y - L)^2) from the call that has part bit 1."""
    lib = _lib.load()
    _require_gpu(L, io_ok=True)
    if L.dtype != Y.dtype:
        raise RuntimeError("windgnn_amd: labels are %s but Y is %s (one I/O type per call)" % (L.dtype, Y.dtype))
    if L.numel() != Y.numel():
        raise RuntimeError("windgnn_amd: MSE operands differ in size: %s vs %s" % (tuple(Y.shape), tuple(L.shape)))
    _require_contiguous(X=X, Y=Y, labels=L, **{"grads[%d]" % i: q for i, q in enumerate(grads)},
                        **{"params[%d]" % i: q for i, q in enumerate(params)})
    A = getattr(A, "blob", A)
    _require_contiguous(adj_matrix=A)
    ws_bytes = lib.wgnn_workspace_bytes(C.byref(d))
    ws = _Workspace.get(X.device, ws_bytes)
    ps = _params_struct(_lib.Params, params, prepared)
    gs = _params_struct(_lib.Grads, grads)
    rc = lib.wgnn_bwd_mse_part(C.byref(d), _ptr(A), _ptr(X), C.byref(ps), _ptr(Y), _ptr(L), grad_scale,
                               _ptr(loss), _ptr(stash), C.byref(gs), _ptr(ws), ws_bytes, _stream(), part)
    _lib.check(rc, "wgnn_bwd_mse_part(%d)" % part)


def prepared_weights(d, params, device):
    """A fresh `prepared` buffer (wgnn_params.prepared) holding the staged images of params' W_ih / b_ih, or None when
    this configuration has none (wgnn_prepared_bytes == 0)."""
    lib = _lib.load()
    nbytes = lib.wgnn_prepared_bytes(C.byref(d))
    if nbytes == 0:
        return None
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    refresh_prepared(d, params, buf)
    return buf


def refresh_prepared(d, params, prepared) -> None:
    """wgnn_prepare_weights: rebuild the images after the caller changed W_ih / b_ih itself (load_state_dict, ...)."""
    lib = _lib.load()
    ws = _Workspace.get(prepared.device, _lib.STATUS_BYTES)
    ps = _params_struct(_lib.Params, params, prepared)
    _lib.check(lib.wgnn_prepare_weights(C.byref(d), C.byref(ps), _ptr(ws), ws.numel(), _stream()), "wgnn_prepare_weights")


def finish_step(d, params, grads, which: int, adam=None, prepared=None, device=None) -> None:
    """wgnn_finish: reduce the deferred partial sums of the backward parts in `which` (4: GRU, 2: conv) into `grads` and,
    with adam = dict(exp_avg=[8 tensors], exp_avg_sq=[8 tensors], step, lr, beta1, beta2, eps), apply Adam to `params`
    in place (and refresh `prepared`) -- one launch.  `which` = _lib.FINISH_ADAM_GRU / FINISH_ADAM_CONV (with adam, no
    reduce bit): the optimiser step of that tensor family only."""
    lib = _lib.load()
    ws_bytes = lib.wgnn_workspace_bytes(C.byref(d))
    ws = _Workspace.get(device if device is not None else grads[0].device, ws_bytes)
    ps = _params_struct(_lib.Params, params, prepared)
    gs = _params_struct(_lib.Grads, grads)
    ad = None
    if adam is not None:
        ad = _lib.Adam()
        ad.exp_avg = _params_struct(_lib.Grads, adam["exp_avg"])
        ad.exp_avg_sq = _params_struct(_lib.Grads, adam["exp_avg_sq"])
        ad.step, ad.lr, ad.beta1, ad.beta2, ad.eps = adam["step"], adam["lr"], adam["beta1"], adam["beta2"], adam["eps"]
    rc = lib.wgnn_finish(C.byref(d), C.byref(ps), C.byref(gs), which, C.byref(ad) if ad is not None else None, _ptr(ws),
                         ws_bytes, _stream())
    _lib.check(rc, "wgnn_finish(%d%s)" % (which, ", adam" if adam is not None else ""))


class GCNGRUFunction(torch.autograd.Function):
    """Y = GCN_GRU(A, X; 8 params).  Gradients flow to the parameters only: the reference's
    adjacency and inputs do not require grad (src/main.py:26, src/step4_sequence_preparer.py:58)."""

    @staticmethod
    def forward(ctx, A, X, math, *params):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[0]:
            # the reference's autograd would propagate into attr_matrix / adj_matrix; this path has no dX / dA
            # (neither requires grad in the reference loop) and must not drop a requested gradient silently
            raise RuntimeError("windgnn_amd: GCN_GRU gives gradients for its 8 parameters only; attr_matrix / "
                               "adj_matrix with requires_grad=True are not supported (detach them, or stack "
                               "GraphConvLayer modules, which do return dX)")
        X = X.contiguous()
        params = tuple(p.contiguous() for p in params)
        need = any(ctx.needs_input_grad[3:])
        Y, stash, d = gcn_gru_forward_raw(A, X, params, math, want_stash=need)
        ctx.d = d
        ctx.stash = stash
        ctx.save_for_backward(_adj(A)[0], X, Y, *params)   # dense [S,S] or the CSR blob; d.adj_format says which
        return Y

    @staticmethod
    def backward(ctx, dY):
        A, X, Y, *params = ctx.saved_tensors
        if ctx.stash is None:
            raise RuntimeError("windgnn_amd: backward called but the forward ran without a stash")
        sizes = [p.numel() for p in params]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=X.device)
        grads = [g.view_as(p) for g, p in zip(flat.split(sizes), params)]
        gcn_gru_backward_raw(ctx.d, A, X, params, Y, dY.float().contiguous(), ctx.stash, grads)   # dY is always fp32
        return (None, None, None, *grads)


def gcn_gru(A, X, params, math=_lib.MATH_F32):
    return GCNGRUFunction.apply(A, X, math, *params)


class GraphConvFunction(torch.autograd.Function):
    """out = relu(A X W + b) for X [..., S, F_in], W [F_in, F_out] (src/step5_gcn_layer_model.py:13-23).  Dense adjacency: any
    widths up to 64 (13 -> 13, the reference model's own, on the MFMA kernels); CSR adjacency: 13 -> 13 only."""

    @staticmethod
    def forward(ctx, A, X, W, b):
        lib = _lib.load()
        _require_gpu(X, W, b)
        X, W, b = X.contiguous(), W.contiguous(), b.contiguous()
        S, F = X.shape[-2], X.shape[-1]
        if W.dim() != 2 or W.shape[0] != F or b.shape != (W.shape[1],):
            # the reference's torch.matmul(adj_attr, self.weight) would fail the same way (step5:18)
            raise RuntimeError("GraphConvLayer: attr_matrix has %d features but the weight is %s and the bias %s"
                               % (F, tuple(W.shape), tuple(b.shape)))
        Fo = W.shape[1]
        A, fmt, nnz = _adj(A, S)
        nt = X.numel() // (S * F)
        out = torch.empty(*X.shape[:-1], Fo, dtype=X.dtype, device=X.device)
        if fmt == _lib.ADJ_CSR:
            if (F, Fo) != (13, 13):
                raise RuntimeError("windgnn_amd: a GraphConvLayer over a CSR adjacency is built for the reference's 13 -> 13 "
                                   "layers only (got %d -> %d); other widths need a dense adjacency (S <= 64)" % (F, Fo))
            rc = lib.wgnn_gcn_layer_csr_fwd(nt, S, F, nnz, _ptr(A), _ptr(X), _ptr(W), _ptr(b), _ptr(out), _stream())
        else:
            rc = lib.wgnn_gcn_layer_fwd(nt, S, F, Fo, _ptr(A), _ptr(X), _ptr(W), _ptr(b), _ptr(out), _stream())
        _lib.check(rc, "wgnn_gcn_layer_fwd(S=%d, %d -> %d)" % (S, F, Fo))
        ctx.save_for_backward(A, X, W, out)
        ctx.dims = (nt, S, F, Fo, fmt, nnz)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        A, X, W, out = ctx.saved_tensors
        nt, S, F, Fo, fmt, nnz = ctx.dims
        dout = dout.contiguous()
        dW = torch.empty_like(W)
        db = torch.empty(Fo, dtype=torch.float32, device=X.device)
        dX = torch.empty_like(X) if ctx.needs_input_grad[1] else None
        if fmt == _lib.ADJ_CSR:
            nbytes = lib.wgnn_gcn_layer_csr_workspace_bytes(nt, S, F)
            wsp, _ = _scratch(X.device, nbytes)
            rc = lib.wgnn_gcn_layer_csr_bwd(nt, S, F, nnz, _ptr(A), _ptr(X), _ptr(W), _ptr(out), _ptr(dout), _ptr(dW),
                                            _ptr(db), _ptr(dX), wsp, nbytes, _stream())
            _lib.check(rc, "wgnn_gcn_layer_csr_bwd")
            return None, dX, dW, db
        nbytes = lib.wgnn_gcn_layer_workspace_bytes(nt, S, F, Fo)
        wsp, _ = _scratch(X.device, nbytes)
        rc = lib.wgnn_gcn_layer_bwd(nt, S, F, Fo, _ptr(A), _ptr(X), _ptr(W), _ptr(out), _ptr(dout), _ptr(dW), _ptr(db),
                                    _ptr(dX), wsp, nbytes, _stream())
        _lib.check(rc, "wgnn_gcn_layer_bwd")
        return None, dX, dW, db


class GRUFunction(torch.autograd.Function):
    """Y [B,T,H] = nn.GRU(I, H, batch_first=True)(g [B,T,I]) with h0 = 0 (src/step6_gcn_gru_combined_model.py:23), through
    wgnn_gru_fwd / wgnn_gru_bwd: gradients for g and the four GRU tensors.  What GCN_GRU runs behind two GraphConvLayers when its
    input_dim / hidden_dim are not the reference model's 13 (exact fp32; I must be S * 13: step6:16)."""

    @staticmethod
    def forward(ctx, g, S, w_ih, w_hh, b_ih, b_hh):
        lib = _lib.load()
        _require_gpu(g, w_ih, w_hh, b_ih, b_hh)
        g = g.contiguous()
        params = [p.contiguous() for p in (w_ih, w_hh, b_ih, b_hh)]
        B, T, I = g.shape
        H = w_hh.shape[1]
        d = _lib.Dims(B, T, S, 13, H, _lib.MATH_F32, _lib.ADJ_DENSE, 0, _lib.IO_F32)
        ws_bytes = lib.wgnn_workspace_bytes(C.byref(d))
        if ws_bytes == 0 or I != S * 13:
            _lib.check(-5 if I == S * 13 else -2, "wgnn_gru_fwd(B=%d,T=%d,S=%d,I=%d,H=%d)" % (B, T, S, I, H))
        ws = _Workspace.get(g.device, ws_bytes)
        need = any(ctx.needs_input_grad)
        stash = torch.empty(lib.wgnn_stash_bytes(C.byref(d)), dtype=torch.uint8, device=g.device) if need else None
        Y = torch.empty(B, T, H, dtype=torch.float32, device=g.device)
        ps = _lib.Params()
        ps.w_ih, ps.w_hh, ps.b_ih, ps.b_hh = (q.data_ptr() for q in params)
        _lib.check(lib.wgnn_gru_fwd(C.byref(d), _ptr(g), C.byref(ps), _ptr(Y), _ptr(stash), _ptr(ws), ws_bytes, _stream()),
                   "wgnn_gru_fwd")
        ctx.d, ctx.stash = d, stash
        ctx.save_for_backward(g, Y, *params)
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.load()
        g, Y, *params = ctx.saved_tensors
        grads = [torch.empty_like(q) for q in params]
        dg = torch.empty_like(g)
        ws_bytes = lib.wgnn_workspace_bytes(C.byref(ctx.d))
        ws = _Workspace.get(g.device, ws_bytes)
        ps, gs = _lib.Params(), _lib.Grads()
        ps.w_ih, ps.w_hh, ps.b_ih, ps.b_hh = (q.data_ptr() for q in params)
        gs.w_ih, gs.w_hh, gs.b_ih, gs.b_hh = (q.data_ptr() for q in grads)
        _lib.check(lib.wgnn_gru_bwd(C.byref(ctx.d), _ptr(g), C.byref(ps), _ptr(Y), _ptr(dY.float().contiguous()),
                                    _ptr(ctx.stash), C.byref(gs), _ptr(dg), _ptr(ws), ws_bytes, _stream()), "wgnn_gru_bwd")
        return (dg, None, *grads)


def mse_loss_grad(Y, L, grad_scale: float = 1.0, want_grad=True):
    """loss (0-dim tensor on device) and dY for nn.MSELoss()(Y, L) (src/main.py:49,72)."""
    lib = _lib.load()
    _require_gpu(Y, L)
    Y, L = Y.contiguous(), L.contiguous()
    if Y.numel() != L.numel():
        raise RuntimeError("windgnn_amd: MSE operands differ in size: %s vs %s" % (tuple(Y.shape), tuple(L.shape)))
    dY = torch.empty_like(Y) if want_grad else None
    loss = torch.empty((), dtype=torch.float32, device=Y.device)
    wsp, wsn = _scratch(Y.device, 4096)
    rc = lib.wgnn_mse_loss_grad(_ptr(Y), _ptr(L), Y.numel(), grad_scale, _ptr(dY), _ptr(loss), wsp, wsn, _stream())
    _lib.check(rc, "wgnn_mse_loss_grad")
    return loss, dY


def adam_step_(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    lib = _lib.load()
    _require_gpu(param, grad, exp_avg, exp_avg_sq)
    rc = lib.wgnn_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), step, lr, beta1,
                            beta2, eps, _stream())
    _lib.check(rc, "wgnn_adam_step")
