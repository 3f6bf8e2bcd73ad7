"""Host-side mirror of the reference's training-loop body (src/main.py:64-80) on flat buffers:
forward -> MSE + backward (wgnn_bwd_mse_part) -> (data-parallel all-reduce) -> Adam, every op a C-ABI call.

Data parallel: windows are independent, so each rank runs its own shard of
windows; the 8 gradients live in ONE flat fp32 bucket (167 440 floats at S=34).  Per step the bucket is summed by
TWO all-reduces on RCCL's stream: the GRU gradients (99.8 % of the bytes) as soon as the weight-gradient GEMMs are
done, overlapped with the rest of the backward (dg GEMM + GCN backward), then the 364 conv gradients together with
the scalar loss.  dY is pre-scaled by n_local / n_global (= 1 / world_size for equal shards) so the summed bucket
equals the gradient of the big-batch mean loss, and the returned loss is the big-batch mean loss on every rank."""
from __future__ import annotations

import torch

from . import _lib
from .functional import adam_step_, check_range_status, gcn_gru_backward_mse_raw, gcn_gru_forward_raw
from .modules import GCN_GRU


class TrainStep:
    def __init__(self, model: GCN_GRU, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, check_every: int = 100):
        self.model = model
        self.params = list(model.hot_path_parameters())
        sizes = [p.numel() for p in self.params]
        dev = self.params[0].device
        self.flat_p = torch.cat([p.detach().reshape(-1) for p in self.params]).contiguous()
        # gradient bucket with a 4-float header (16-byte aligned bucket): header[3] = the step's loss, so that the loss
        # rides in the conv-gradient all-reduce (the conv gradients are the first 364 floats of the bucket)
        self._gbuf = torch.zeros(self.flat_p.numel() + 4, dtype=torch.float32, device=dev)
        self.flat_g = self._gbuf[4:]
        self._loss = self._gbuf[3]
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.p_views, self.g_views = [], []
        for p, pv, gv in zip(self.params, self.flat_p.split(sizes), self.flat_g.split(sizes)):
            p.data = pv.view_as(p)              # parameters become views of the flat bucket
            p.grad = gv.view_as(p)
            self.p_views.append(p.data)
            self.g_views.append(p.grad)
        self.n_conv = sum(sizes[:4])
        self.lr, self.betas, self.eps = lr, betas, eps
        self.steps = 0
        self.group = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # an explicitly passed group runs the collective path even with one rank (the all-reduces execute)
        self.collective = self.world > 1 or process_group is not None
        self._shard = {}                        # local window count -> (grad scale, loss weight) of this job
        self.check_every = check_every          # f16x3 / f16: read the library's range-status word every N steps
        self.device = dev

    def _scales(self, n_local: int):
        """(n_local / n_global, equal_shards) for this local window count; one tiny all-reduce the first time a
        count is seen (shards may differ by one window: distributed.shard_range)."""
        if n_local not in self._shard:
            t = torch.tensor([float(n_local), float(n_local), -float(n_local)], device=self.device, dtype=torch.float64)
            tot = t[:1].clone()
            torch.distributed.all_reduce(tot, group=self.group)
            ext = t[1:].clone()
            torch.distributed.all_reduce(ext, op=torch.distributed.ReduceOp.MAX, group=self.group)
            self._shard[n_local] = (n_local / float(tot.item()), float(ext[0].item()) == -float(ext[1].item()))
        return self._shard[n_local]

    def forward_backward(self, A, X, L):
        """src/main.py:66,72,79: returns (loss, Y); gradients land in the flat bucket."""
        Y, stash, d = gcn_gru_forward_raw(A, X, self.p_views, self.model.math, want_stash=True, labels=L)
        loss = self._loss
        gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, 1.0, part=7 | 8)
        return loss, Y

    def step(self, A, X, L):
        if self.collective:
            # Overlap: the GRU gradients (99.8 % of the bucket) are final after parts 1|4 of the backward, so
            # their all-reduce runs on RCCL's stream while part 2 (dg GEMM + GCN backward, ~30 % of the
            # step) still computes; the 364 conv gradients and the loss follow in a second, tiny all-reduce.
            gs, equal = self._scales(X.shape[0])
            Y, stash, d = gcn_gru_forward_raw(A, X, self.p_views, self.model.math, want_stash=True, labels=L)
            loss = self._loss
            gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, gs, part=1 | 4 | 8)
            if not equal:
                loss.mul_(gs * self.world)      # weight of this shard's mean in the global mean, times world
            work = torch.distributed.all_reduce(self.flat_g[self.n_conv:], group=self.group, async_op=True)
            gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, gs, part=2)
            torch.distributed.all_reduce(self._gbuf[3:4 + self.n_conv], group=self.group)
            work.wait()
            loss = loss / self.world            # sum of the shard means (weighted if unequal) -> big-batch mean
        else:
            loss, Y = self.forward_backward(A, X, L)
            loss = loss.clone()
        self.steps += 1
        adam_step_(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.steps, self.lr,
                   self.betas[0], self.betas[1], self.eps)                   # src/main.py:80
        if self.check_every and self.model.math != _lib.MATH_F32 and self.steps % self.check_every == 0:
            self.check()
        return loss, Y

    def check(self):
        """Raise if a kernel of the fp16-plane modes reported a value outside fp16's range (one 4-byte read)."""
        check_range_status(self.device)
