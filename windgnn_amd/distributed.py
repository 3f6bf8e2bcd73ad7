"""Data-parallel host logic.  Windows are independent (h0 = 0 per window,
src/step6_gcn_gru_combined_model.py:23 passes no h0), so the path shards over B with no data-path
collective; the only exchange is the sum of the flat gradient bucket (`BucketExchange`): by default ONE all-reduce of
[loss | conv | GRU gradients] per step (`all_reduce_all`); the two-collective form (`start_gru` / `start_conv` / `finish`: the
GRU gradients as soon as they are final, overlapped with the rest of the backward, then the conv gradients with the loss) is
kept and tested, but its cross-stream dependencies cost more than the overlap saves (trainer.py).

Backend-agnostic (nccl == RCCL over xGMI on the GPU box, gloo in the CPU tests).  `TrainStep` (trainer.py) and
`bench.py` run exactly this class; tests/test_distributed_cpu.py runs it with the oracle standing in for the kernels.

RCCL on this pool needs dmabuf IPC: `ensure_rccl_env()` exports HSA_ENABLE_IPC_MODE_LEGACY=0 and must run before the
process initialises the GPU (bench.py calls it first thing)."""
from __future__ import annotations

import os
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist

HEADER = 4                  # floats in front of the gradients in the bucket (16-byte aligned); word 3 = the loss
LOSS_SLOT = 3


def ensure_rccl_env() -> None:
    """The host driver of this pool only supports dmabuf IPC; with the legacy mode RCCL's (and torch's) cross-process
    buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.  Must be set before the first HIP call."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def shard_range(n_windows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; the first n % world ranks get one extra window."""
    base, rem = divmod(n_windows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_windows(X: torch.Tensor, L: torch.Tensor, rank: int, world: int):
    """This rank's windows, as contiguous tensors (the raw entry points refuse strided views)."""
    lo, hi = shard_range(X.shape[0], rank, world)
    return X[lo:hi].contiguous(), L[lo:hi].contiguous()


def grad_scale_for_shard(n_local: int, n_global: int) -> float:
    """dY scale that makes SUM over ranks of shard gradients equal the gradient of the global
    mean loss: local mean-loss gradient times n_local / n_global (== 1/world for equal shards)."""
    return float(n_local) / float(n_global)


def flatten(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for t in tensors])


class DirectRccl:
    """One RCCL communicator of our own, driven through ctypes, so that the gradient all-reduce is enqueued ON THE CALLER'S
    STREAM: ncclAllReduce(bucket, bucket, n, ncclFloat32, ncclSum, comm, current stream) sits between wgnn_finish(6) and
    wgnn_finish(0, adam) like any other kernel of the step.  torch.distributed's NCCL backend runs every collective on a
    stream of its own and brackets it with two event dependencies; on this platform each such hop idles the queue for
    ~10-20 us (profiles/r3_collective_path_one_rank.txt), i.e. more than the 0.67 MB all-reduce itself.

    Bootstrap (the canonical one): rank 0 draws ncclGetUniqueId, the 128 bytes travel through the EXISTING torch process
    group (one broadcast at construction), every rank calls ncclCommInitRank.  The library is the librccl.so torch itself
    loaded (no second copy in the process).  Whether the direct path is used is decided COLLECTIVELY (a MIN all-reduce of
    "I could load the library and my backend is nccl"): either every rank takes it or none does."""

    _FLOAT32, _SUM = 7, 0                      # ncclDataType_t ncclFloat32, ncclRedOp_t ncclSum

    def __init__(self, device: torch.device, group=None):
        import ctypes as C
        self.comm = None
        self._lib = None
        ok = 0
        lib = None
        try:
            if dist.get_backend(group) == "nccl":
                lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
                for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy", "ncclGetErrorString"):
                    getattr(lib, name)
                ok = 1
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)          # every rank, whatever its own answer
        if int(flag.item()) == 0:
            return

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_byte * 128)]

        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = UniqueId()
        rc0 = lib.ncclGetUniqueId(C.byref(uid)) if rank == 0 else 0
        t = torch.tensor(list(bytes(uid)), dtype=torch.uint8).to(device)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(t.cpu().tolist())
        C.memmove(C.byref(uid), raw, 128)
        comm = C.c_void_p()
        torch.cuda.set_device(device)
        rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank) if rc0 == 0 else rc0
        good = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=device)
        dist.all_reduce(good, op=dist.ReduceOp.MIN, group=group)          # again collectively: all or none
        if int(good.item()) == 0:
            if rc == 0:
                lib.ncclCommDestroy(comm)
            return
        self._lib, self.comm = lib, comm
        # known-answer check of the new communicator before any gradient goes through it: [1, rank + 1] must sum to
        # [world, world (world + 1) / 2] on every rank, else every rank drops back to torch.distributed's own collective
        probe = torch.tensor([1.0, float(rank + 1), 0.0, 0.0], dtype=torch.float32, device=device)
        try:
            self.all_reduce_(probe)
            got = probe.cpu().tolist()
            fine = int(got[0] == float(world) and got[1] == world * (world + 1) / 2.0)
        except RuntimeError:
            fine = 0
        good.fill_(fine)
        dist.all_reduce(good, op=dist.ReduceOp.MIN, group=group)
        if int(good.item()) == 0:
            lib.ncclCommDestroy(comm)
            self._lib, self.comm = None, None

    def all_reduce_(self, t: torch.Tensor) -> None:
        """In-place fp32 sum over the ranks, enqueued on the current stream of t's device."""
        rc = self._lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), self._FLOAT32, self._SUM, self.comm,
                                     torch.cuda.current_stream(t.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("windgnn_amd: ncclAllReduce failed: %s" % self._lib.ncclGetErrorString(rc).decode())

    def close(self) -> None:
        if self.comm is not None:
            self._lib.ncclCommDestroy(self.comm)
            self.comm = None


class BucketExchange:
    """The collectives of one data-parallel step over a flat bucket `[4-float header | conv grads | GRU grads]`.

    Every collective here is issued unconditionally by every rank in the same order, whatever the rank-local state
    (a cache keyed by the local window count let one rank skip a collective another one issued).
    The global window count either comes from the caller (`n_global`, e.g. bench.py's fixed world * B: no extra
    collective, no host sync) or is all-reduced on EVERY call."""

    def __init__(self, bucket: torch.Tensor, n_conv: int, group=None, direct=None):
        self.bucket = bucket
        self.n_conv = n_conv
        self.group = group
        self.world = dist.get_world_size(group)
        # the single all-reduce on the compute stream through our own RCCL communicator (GPU buckets on the nccl backend;
        # WGNN_RCCL_DIRECT=0 or direct=False keeps torch.distributed's own stream); the decision is taken collectively
        if direct is None:
            direct = os.environ.get("WGNN_RCCL_DIRECT", "1") != "0"
        self.direct = None
        if direct and bucket.is_cuda:
            d = DirectRccl(bucket.device, group)
            if d.comm is not None:
                self.direct = d

    def shard_weight(self, n_local: int, n_global: Optional[int] = None) -> float:
        """n_local / n_global: the dY scale of this shard and the weight of its mean loss in the global mean."""
        if n_global is None:
            t = torch.tensor([float(n_local)], dtype=torch.float64, device=self.bucket.device)
            dist.all_reduce(t, group=self.group)            # every step, every rank
            n_global = int(round(float(t.item())))
        # n_local == 0 is legal (fewer windows than ranks): that rank contributes a zero bucket and still issues every
        # collective (TrainStep._empty_shard_step); what is refused is refused identically on every rank or is a local bug
        if n_local < 0 or n_global < n_local or n_global < 1:
            raise RuntimeError("windgnn_amd: shard of %d windows in a global batch of %d" % (n_local, n_global))
        return grad_scale_for_shard(n_local, n_global)

    def all_reduce_all(self, weight: float) -> None:
        """Weight this shard's mean loss and sum [loss | conv grads | GRU grads] in ONE all-reduce (the current stream
        waits for it; no host sync).  What TrainStep runs by default: see trainer.py for the measurement behind it."""
        if weight != 1.0:
            self.bucket[LOSS_SLOT].mul_(weight)
        if self.direct is not None:
            self.direct.all_reduce_(self.bucket)             # on the compute stream: no stream hop around 0.67 MB
        else:
            dist.all_reduce(self.bucket, group=self.group)   # the whole, 16-byte aligned bucket (header words 0-2 are zeros)

    def start_gru(self):
        """Sum the GRU gradients (final after backward parts 1|4); returns the async work handle."""
        return dist.all_reduce(self.bucket[HEADER + self.n_conv:], group=self.group, async_op=True)

    def start_conv(self, weight: float):
        """Weight this shard's mean loss and start the small all-reduce of [loss | conv grads] (final after backward part 2);
        returns the async work handle.  The caller may run the GRU tensors' optimiser step under it."""
        if weight != 1.0:
            self.bucket[LOSS_SLOT].mul_(weight)
        return dist.all_reduce(self.bucket[LOSS_SLOT:HEADER + self.n_conv], group=self.group, async_op=True)

    def finish(self, work, weight: float) -> None:
        """Weight this shard's mean loss, sum the conv gradients + the loss in one small all-reduce, join `work`:
        afterwards the bucket holds the big-batch gradient and word 3 the big-batch mean loss on every rank."""
        if weight != 1.0:
            self.bucket[LOSS_SLOT].mul_(weight)
        dist.all_reduce(self.bucket[LOSS_SLOT:HEADER + self.n_conv], group=self.group)
        work.wait()
