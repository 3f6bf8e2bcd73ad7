"""Data-parallel helpers.  Windows are independent (h0 = 0 per window,
src/step6_gcn_gru_combined_model.py:23 passes no h0), so the path shards over B with no data-path
collective; the only exchange is ONE all-reduce of the flat gradient bucket per step.

These helpers are backend-agnostic host logic (nccl == RCCL on the GPU box, gloo in CPU tests)."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_windows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; the first n % world ranks get one extra window."""
    base, rem = divmod(n_windows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_windows(X: torch.Tensor, L: torch.Tensor, rank: int, world: int):
    lo, hi = shard_range(X.shape[0], rank, world)
    return X[lo:hi], L[lo:hi]


def grad_scale_for_shard(n_local: int, n_global: int) -> float:
    """dY scale that makes SUM over ranks of shard gradients equal the gradient of the global
    mean loss: local mean-loss gradient times n_local / n_global (== 1/world for equal shards)."""
    return float(n_local) / float(n_global)


def allreduce_flat_(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """One sum all-reduce of the single flat gradient bucket (167 440 fp32 at S=34)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def flatten(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for t in tensors])
