"""N2 / N4 host wrappers: on-device window batcher and evaluation read-out (C-ABI kernels)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from .functional import _ptr, _require_gpu, _stream

LABEL_FEATURE = 11     # reference column 13 ("Wind Speed 10 m Avg.") minus the 2 id columns


def make_windows(feat: torch.Tensor, seq_len: int, starts: Optional[Sequence[int]] = None, n_windows: Optional[int] = None):
    """feat [Ttot, S, 13] on the GPU -> (X [B,seq,S,13], L [B,seq,3S]) as src/step4_sequence_preparer.py:7-21.
    Default: the reference's non-overlapping windows i*seq_len; pass `starts` (e.g. a permutation, :23-26)
    to choose them."""
    lib = _lib.load()
    _require_gpu(feat)
    feat = feat.contiguous()
    Ttot, S, F = feat.shape
    if starts is None:
        # the reference's window count, len(data) // seq_length (:10); when the last window's +3 h labels do not
        # fit (len % seq_length < 3) the reference fails on a ragged concatenate and this call raises too
        B = n_windows if n_windows is not None else Ttot // seq_len
        host, dev = None, None
    else:
        B = len(starts)
        host = (C.c_int32 * B)(*[int(s) for s in starts])
        dev = torch.tensor(list(starts), dtype=torch.int32, device=feat.device)
    if B < 1:
        raise RuntimeError("windgnn_amd.make_windows: no complete window of %d (+3 label) steps in %d rows" % (seq_len, Ttot))
    X = torch.empty(B, seq_len, S, F, dtype=torch.float32, device=feat.device)
    L = torch.empty(B, seq_len, 3 * S, dtype=torch.float32, device=feat.device)
    rc = lib.wgnn_make_windows(_ptr(feat), Ttot, S, F, seq_len, LABEL_FEATURE, host, _ptr(dev), B, _ptr(X), _ptr(L), _stream())
    _lib.check(rc, "wgnn_make_windows")
    return X, L


def predict_last(Y: torch.Tensor, wind_min: float, wind_max: float) -> torch.Tensor:
    """De-normalised last-timestep rows [B, 3S] (columns [0:S] = +1 h, [S:2S] = +2 h, [2S:3S] = +3 h):
    src/main.py:103,116,131,146."""
    lib = _lib.load()
    _require_gpu(Y)
    if Y.dim() == 2:
        Y = Y.unsqueeze(0)
    Y = Y.contiguous()
    B, T, H = Y.shape
    out = torch.empty(B, H, dtype=torch.float32, device=Y.device)
    _lib.check(lib.wgnn_predict_last(_ptr(Y), B, T, H, float(wind_min), float(wind_max), _ptr(out), _stream()),
               "wgnn_predict_last")
    return out


def forward_last(model, adj_matrix, attr_matrix: torch.Tensor, wind_min: float, wind_max: float) -> torch.Tensor:
    """The evaluation loop's `model(adj, x)` + last-row de-normalisation (src/main.py:100-103,116) as one C-ABI call,
    wgnn_fwd_last: [B, 3S] (B = 1: [3S]) without materialising the other T-1 output rows."""
    from .functional import _Workspace, _adj, _params_struct
    lib = _lib.load()
    _require_gpu(attr_matrix)
    if not getattr(model, "fused", True):        # other widths than the reference model's 13 / 13: the module's own forward
        with torch.no_grad():
            out = model(adj_matrix, attr_matrix)
        return predict_last(out if out.dim() == 3 else out.unsqueeze(0), wind_min, wind_max).squeeze(0)
    X = attr_matrix.contiguous()
    B, T, S, F = X.shape
    A, fmt, nnz = _adj(adj_matrix, S)
    params = [p.detach() for p in model.hot_path_parameters()]
    H = params[5].shape[1]
    d = _lib.Dims(B, T, S, F, H, model.math, fmt, nnz, _lib.IO_F32)
    nbytes = lib.wgnn_workspace_bytes(C.byref(d))
    if nbytes == 0:
        _lib.check(-5, "wgnn_workspace_bytes")
    ws = _Workspace.get(X.device, nbytes)
    out = torch.empty(B, H, dtype=torch.float32, device=X.device)
    ps = _params_struct(_lib.Params, params)
    _lib.check(lib.wgnn_fwd_last(C.byref(d), _ptr(A), _ptr(X), C.byref(ps), float(wind_min), float(wind_max), _ptr(out),
                                 _ptr(ws), nbytes, _stream()), "wgnn_fwd_last")
    return out.squeeze(0)
