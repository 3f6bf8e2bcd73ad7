"""Host-side mirror of the reference's training-loop body (src/main.py:64-80) on flat buffers:
forward -> MSE + backward (wgnn_bwd_mse_part) -> (data-parallel all-reduce) -> Adam, every op a C-ABI call.

Data parallel: windows are independent, so each rank runs its own shard of
windows; the 8 gradients live in ONE flat fp32 bucket (167 440 floats at S=34) that is summed
with a single RCCL all-reduce per step.  dY is pre-scaled by 1/world_size so the summed bucket
equals the gradient of the big-batch mean loss."""
from __future__ import annotations

import torch

from .distributed import allreduce_flat_
from .functional import adam_step_, gcn_gru_backward_mse_raw, gcn_gru_forward_raw
from .modules import GCN_GRU


class TrainStep:
    def __init__(self, model: GCN_GRU, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None):
        self.model = model
        self.params = list(model.hot_path_parameters())
        sizes = [p.numel() for p in self.params]
        dev = self.params[0].device
        self.flat_p = torch.cat([p.detach().reshape(-1) for p in self.params]).contiguous()
        self.flat_g = torch.zeros_like(self.flat_p)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.p_views, self.g_views = [], []
        for p, pv, gv in zip(self.params, self.flat_p.split(sizes), self.flat_g.split(sizes)):
            p.data = pv.view_as(p)              # parameters become views of the flat bucket
            p.grad = gv.view_as(p)
            self.p_views.append(p.data)
            self.g_views.append(p.grad)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.steps = 0
        self.group = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.device = dev

    def forward_backward(self, A, X, L):
        """src/main.py:66,72,79: returns (loss, Y); gradients land in the flat bucket."""
        Y, stash, d = gcn_gru_forward_raw(A, X, self.p_views, self.model.math, want_stash=True)
        loss = torch.empty((), dtype=torch.float32, device=Y.device)
        gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, 1.0 / self.world)
        return loss, Y

    def step(self, A, X, L):
        if self.world > 1:
            # Overlap: the GRU gradients (99.8 % of the bucket) are final after part 1 of the backward, so
            # their all-reduce runs on RCCL's stream while part 2 (dg GEMM + GCN backward, ~30 % of the
            # step) still computes; the 364 conv gradients follow in a second, tiny all-reduce.
            Y, stash, d = gcn_gru_forward_raw(A, X, self.p_views, self.model.math, want_stash=True)
            loss = torch.empty((), dtype=torch.float32, device=Y.device)
            gs = 1.0 / self.world
            gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, gs, part=1 | 4)
            n_conv = sum(g.numel() for g in self.g_views[:4])
            work = torch.distributed.all_reduce(self.flat_g[n_conv:], group=self.group, async_op=True)
            gcn_gru_backward_mse_raw(d, A, X, self.p_views, Y, L, stash, self.g_views, loss, gs, part=2)
            allreduce_flat_(self.flat_g[:n_conv], self.group)
            work.wait()
        else:
            loss, Y = self.forward_backward(A, X, L)
        self.steps += 1
        adam_step_(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.steps, self.lr,
                   self.betas[0], self.betas[1], self.eps)                   # src/main.py:80
        return loss, Y
