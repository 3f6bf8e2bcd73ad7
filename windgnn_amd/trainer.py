"""Host-side mirror of the reference's training-loop body (src/main.py:64-80) on flat buffers:
forward -> MSE + backward (wgnn_bwd_mse_part) -> (data-parallel all-reduce) -> Adam, every op a C-ABI call.

Data parallel: windows are independent, so each rank runs its own shard of
windows; the 8 gradients live in ONE flat fp32 bucket (167 440 floats at S=34).  Per step the bucket is summed by ONE
all-reduce of [loss | conv gradients | GRU gradients] (0.67 MB) between the reduce-only wgnn_finish(6) and the optimiser's
wgnn_finish(0, adam).  (Round 3 measured the alternative it replaced -- two all-reduces, the GRU gradients' one started
asynchronously after the weight-gradient GEMMs and overlapped with the dg GEMM + GCN backward -- on a one-rank group, where the
collectives themselves cost nothing: +65 us per 632 us step for its cross-stream dependencies and extra launches against +10 us
for this form; `overlap_collectives=True` still selects it.)  dY is pre-scaled by n_local / n_global (= 1 / world_size for
equal shards) so the summed bucket equals the gradient of the big-batch mean loss, and the returned loss is the big-batch mean
loss on every rank.  The collectives themselves live in distributed.BucketExchange (every rank issues every collective on
every step).

The loss a step returns is a 0-dim VIEW of the bucket's header (no clone launch per step): read it (float(loss)) or
`.clone()` it before the next step -- a list of kept losses would all show the latest value.

A rank whose shard is EMPTY (fewer windows than ranks) still issues every collective of the step: it contributes a zero
bucket and runs the optimiser's launch on the summed gradient, so no rank is left waiting in an all-reduce.

A step that RAISES (a refused launch, a failed collective) leaves the optimiser state undefined: `steps` counts completed
steps only, but in the two-collective form the GRU tensors and their moments may already have been stepped when a later
launch of the same step fails, and a retry would then apply the same bias-correction step number to them twice.  Do not
retry a failed step on the same TrainStep: rebuild it from a checkpoint of the parameters (state_dict) instead."""
from __future__ import annotations

import torch

from . import _lib
from .distributed import HEADER, LOSS_SLOT, BucketExchange
from .functional import (check_range_status, finish_step, gcn_gru_backward_mse_raw, gcn_gru_forward_raw,
                         prepared_weights, refresh_prepared)
from .modules import GCN_GRU


class TrainStep:
    def __init__(self, model: GCN_GRU, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, check_every: int = 100, overlap_collectives: bool = False, direct_rccl=None,
                 rccl_loader=None):
        if not getattr(model, "fused", True):
            raise RuntimeError("windgnn_amd: TrainStep drives the fused hot path, i.e. the reference model's own widths "
                               "(input_dim = hidden_dim = 13, src/main.py:41); a GCN_GRU of other widths trains through "
                               "autograd (loss.backward() + torch.optim.Adam, as src/main.py:66-80 does)")
        self.model = model
        self.params = list(model.hot_path_parameters())
        sizes = [p.numel() for p in self.params]
        dev = self.params[0].device
        self.flat_p = torch.cat([p.detach().reshape(-1) for p in self.params]).contiguous()
        # gradient bucket with a 4-float header (16-byte aligned bucket): header[3] = the step's loss, so that the loss
        # rides in the conv-gradient all-reduce (the conv gradients are the first 364 floats of the bucket)
        self._gbuf = torch.zeros(self.flat_p.numel() + HEADER, dtype=torch.float32, device=dev)
        self.flat_g = self._gbuf[HEADER:]
        self._loss = self._gbuf[LOSS_SLOT]
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.p_views, self.g_views = [], []
        for p, pv, gv in zip(self.params, self.flat_p.split(sizes), self.flat_g.split(sizes)):
            p.data = pv.view_as(p)              # parameters become views of the flat bucket
            p.grad = gv.view_as(p)
            self.p_views.append(p.data)
            self.g_views.append(p.grad)
        self.m_views = [t.view_as(p) for t, p in zip(self.exp_avg.split(sizes), self.params)]
        self.v_views = [t.view_as(p) for t, p in zip(self.exp_avg_sq.split(sizes), self.params)]
        # the staged images of W_ih (wgnn_params.prepared): built once, then kept current by wgnn_finish's Adam; rebuilt
        # when someone else wrote the parameters (load_state_dict, p.data.copy_, ...: torch's version counter tells)
        self._prepared, self._prepared_version = None, None
        self.n_conv = sum(sizes[:4])
        self.lr, self.betas, self.eps = lr, betas, eps
        self.steps = 0
        self.group = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # an explicitly passed group runs the collective path even with one rank (the all-reduces execute)
        self.collective = self.world > 1 or process_group is not None
        self.overlap_collectives = overlap_collectives
        # direct_rccl: None = only if WGNN_RCCL_DIRECT=1 (opt-in: distributed.DirectRccl); the two-collective form overlaps
        # through torch.distributed's own stream by design and never takes it
        self.exchange = None
        if self.collective:
            self.exchange = BucketExchange(self._gbuf, self.n_conv, process_group,
                                           False if (overlap_collectives and direct_rccl is None) else direct_rccl,
                                           rccl_loader)
        self.check_every = check_every          # f16x3 / f16: read the library's range-status word every N steps
        self.device = dev

    def _param_version(self):
        """torch's in-place version counters of the parameters: load_state_dict / optimiser-free edits through the
        nn.Parameters or the flat buffer bump them; the library's own kernels do not."""
        return (self.flat_p._version,) + tuple(p._version for p in self.params)

    def refresh(self):
        """Call after writing parameters in a way torch does not count (p.data.copy_(...), raw pointers): the staged
        W_ih images are rebuilt on the next step."""
        self._prepared_version = None

    def _images(self, d):
        if self._prepared is None:
            self._prepared = prepared_weights(d, self.p_views, self.device)
        else:
            refresh_prepared(d, self.p_views, self._prepared)
        self._prepared_version = self._param_version()

    def _adam(self):
        # the step being taken: self.steps counts COMPLETED steps (a step that raises does not advance Adam's bias correction)
        return dict(exp_avg=self.m_views, exp_avg_sq=self.v_views, step=self.steps + 1, lr=self.lr, beta1=self.betas[0],
                    beta2=self.betas[1], eps=self.eps)

    def _empty_shard_step(self, A, X, n_global):
        """This rank has no windows in this step: zero bucket, the same collectives as every other rank, the optimiser's
        launch(es) on the summed gradient.  Returns (big-batch mean loss, empty Y)."""
        if not self.collective:
            raise RuntimeError("windgnn_amd: TrainStep.step needs at least one window (got a batch of %s)" % (tuple(X.shape),))
        from .functional import _IO_OF, _adj
        _, T, S, F = X.shape
        _, fmt, nnz = _adj(A, S)
        H = self.p_views[5].shape[1]
        d = _lib.Dims(1, T, S, F, H, self.model.math, fmt, nnz, _IO_OF[X.dtype])    # sizes the finish launch (B-independent)
        if self._prepared_version != self._param_version():
            self._images(d)
        pre = self._prepared
        gs = self.exchange.shard_weight(0, n_global)                   # 0.0; the count collective, if any, is issued
        self._gbuf.zero_()
        if not self.overlap_collectives:
            self.exchange.all_reduce_all(gs)
            finish_step(d, self.p_views, self.g_views, 0, self._adam(), pre, self.device)
        else:
            work = self.exchange.start_gru()
            wconv = self.exchange.start_conv(gs)
            work.wait()
            adam = self._adam()
            finish_step(d, self.p_views, self.g_views, _lib.FINISH_ADAM_GRU, adam, pre, self.device)
            wconv.wait()
            finish_step(d, self.p_views, self.g_views, _lib.FINISH_ADAM_CONV, adam, pre, self.device)
        self.steps += 1
        return self._loss, torch.empty(0, T, H, dtype=X.dtype, device=X.device)

    def forward_backward(self, A, X, L):
        """src/main.py:66,72,79: returns (loss, Y); gradients land in the flat bucket (no optimiser step).  `loss` is a
        view of the bucket's header word (see the module docstring)."""
        X, L = X.contiguous(), L.contiguous()
        Y, stash, d = self._forward(A, X, L)
        loss = self._loss
        gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, 1.0, part=7 | 8,
                                 prepared=self._prepared)
        return loss, Y

    def _forward(self, A, X, L):
        # the first call sizes and builds the images from the dims of this batch (they depend on S, H, math only)
        if self._prepared_version != self._param_version():
            from .functional import _IO_OF, _adj
            B, T, S, F = X.shape
            _, fmt, nnz = _adj(A, S)
            self._images(_lib.Dims(B, T, S, F, self.p_views[5].shape[1], self.model.math, fmt, nnz, _IO_OF[X.dtype]))
        return gcn_gru_forward_raw(A, X, self.p_views, self.model.math, want_stash=True, labels=L, prepared=self._prepared)

    def step(self, A, X, L, n_global=None):
        """One optimiser step on this rank's windows (src/main.py:66-80).  `n_global`: windows of ALL ranks in this step,
        when the caller knows it (a fixed global batch); None = the exchange all-reduces the count on every step (a host
        sync).  Launch-sized tail: ONE wgnn_finish (reduce the deferred partial sums + Adam + next step's W_ih images);
        with a process group two of them around the one all-reduce.  The returned loss is a VIEW of the gradient bucket's
        header word, overwritten by the next step (module docstring): float() or .clone() it to keep it."""
        X, L = X.contiguous(), L.contiguous()   # a strided batch slice is copied here, never read as if dense
        DEFER = _lib.BWD_DEFER
        if X.shape[0] == 0:
            return self._empty_shard_step(A, X, n_global)
        loss = self._loss
        if self.collective and not self.overlap_collectives:
            # the single-rank schedule, with ONE all-reduce of [loss | conv | GRU gradients] between the reduce-only finish and
            # the optimiser's: one collective, one stream dependency each way per step
            gs = self.exchange.shard_weight(X.shape[0], n_global)
            Y, stash, d = self._forward(A, X, L)
            pre = self._prepared
            for part in (1 | 8, 2, 4):
                gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, gs, part=part | DEFER,
                                         prepared=pre)
            finish_step(d, self.p_views, self.g_views, 6, device=self.device)
            self.exchange.all_reduce_all(gs)    # loss: sum of the weighted shard means = the big-batch mean
            finish_step(d, self.p_views, self.g_views, 0, self._adam(), pre, self.device)      # src/main.py:80
        elif self.collective:
            # Overlap: the GRU gradients (99.8 % of the bucket) are final after parts 1|4 of the backward, so
            # their all-reduce runs on RCCL's stream while part 2 (dg GEMM + GCN backward, ~30 % of the
            # step) still computes; the 364 conv gradients and the loss follow in a second, tiny all-reduce.
            gs = self.exchange.shard_weight(X.shape[0], n_global)
            Y, stash, d = self._forward(A, X, L)
            pre = self._prepared
            gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, gs, part=1 | 4 | 8 | DEFER,
                                     prepared=pre)
            finish_step(d, self.p_views, self.g_views, 4, device=self.device)
            work = self.exchange.start_gru()
            gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, gs, part=2 | DEFER, prepared=pre)
            finish_step(d, self.p_views, self.g_views, 2, device=self.device)
            # the conv gradients' (tiny, latency-bound) all-reduce runs under the GRU tensors' optimiser step
            wconv = self.exchange.start_conv(gs)     # loss: sum of the weighted shard means = the big-batch mean
            work.wait()
            adam = self._adam()
            finish_step(d, self.p_views, self.g_views, _lib.FINISH_ADAM_GRU, adam, pre, self.device)   # src/main.py:80
            wconv.wait()
            finish_step(d, self.p_views, self.g_views, _lib.FINISH_ADAM_CONV, adam, pre, self.device)
        else:
            # One rank: BPTT, then the dg GEMM + GCN backward, and the weight-gradient GEMMs LAST, so that wgnn_finish reads
            # their split-K partial sums (115 MB at B = 4096) while they still sit in the Infinity Cache; deferring them
            # across the dg GEMM and the GCN backward (0.8 GB of traffic) had them come back from HBM (finish 35 us)
            Y, stash, d = self._forward(A, X, L)
            pre = self._prepared
            for part in (1 | 8, 2, 4):     # (the order 1, 4, 2 measured the same, 672-680 us either way: round 3)
                gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, 1.0, part=part | DEFER,
                                         prepared=pre)
            finish_step(d, self.p_views, self.g_views, 6, self._adam(), pre, self.device)              # :79 tail + :80
        self.steps += 1                         # only a step whose launches were all accepted counts
        if self.check_every and self.steps % self.check_every == 0 and (
                self.model.math != _lib.MATH_F32 or (self.exchange is not None and self.exchange.direct is not None)):
            self.check()
        return loss, Y

    def close(self):
        """Release the step's own RCCL communicator, if it has one (before torch.distributed.destroy_process_group).
        Also runs from `with TrainStep(...) as tr:` and, as a last resort, from __del__."""
        ex = getattr(self, "exchange", None)
        if ex is not None and ex.direct is not None:
            ex.direct.close()
            ex.direct = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:           # interpreter shutdown: the library may be gone already
            pass

    def check(self):
        """Raise if a kernel of the fp16-plane modes reported a value outside fp16's range (one 4-byte read), or if the
        step's own RCCL communicator reported an asynchronous error (ncclCommGetAsyncError)."""
        if self.exchange is not None and self.exchange.direct is not None:
            self.exchange.direct.check()
        if self.model.math != _lib.MATH_F32:
            check_range_status(self.device)
