"""Data-parallel host logic.  Windows are independent (h0 = 0 per window,
src/step6_gcn_gru_combined_model.py:23 passes no h0), so the path shards over B with no data-path
collective; the only exchange is the sum of the flat gradient bucket (`BucketExchange`): by default ONE all-reduce of
[loss | conv | GRU gradients] per step (`all_reduce_all`); the two-collective form (`start_gru` / `start_conv` / `finish`: the
GRU gradients as soon as they are final, overlapped with the rest of the backward, then the conv gradients with the loss) is
kept and tested, but its cross-stream dependencies cost more than the overlap saves (trainer.py).
`DirectRccl` (opt-in) puts that one all-reduce on the compute stream through a communicator of the step's own.

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


class RcclApi:
    """The six RCCL entry points DirectRccl needs, behind plain Python methods, so that the bootstrap protocol can be run
    against a stand-in on CPU (tests/test_distributed_cpu.py hands DirectRccl a fake with failing draws, failing inits and
    wrong sums).  This one binds the librccl.so torch itself loaded (no second copy in the process).
    Return convention: ncclResult_t first (0 = ncclSuccess)."""

    _FLOAT32, _SUM = 7, 0                      # ncclDataType_t ncclFloat32, ncclRedOp_t ncclSum

    def __init__(self, path: Optional[str] = None):
        import ctypes as C
        self._C = C
        lib = C.CDLL(path or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_byte * 128)]

        self._UniqueId = UniqueId
        for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy", "ncclCommAbort",
                     "ncclCommGetAsyncError", "ncclGetErrorString"):
            getattr(lib, name)                                   # AttributeError = not usable: the caller falls back
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclCommAbort.argtypes = [C.c_void_p]
        lib.ncclCommGetAsyncError.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._lib = lib

    def get_unique_id(self):
        uid = self._UniqueId()
        rc = self._lib.ncclGetUniqueId(self._C.byref(uid))
        return rc, bytes(uid)

    def comm_init_rank(self, world: int, uid: bytes, rank: int):
        u = self._UniqueId()
        self._C.memmove(self._C.byref(u), uid, 128)
        comm = self._C.c_void_p()
        rc = self._lib.ncclCommInitRank(self._C.byref(comm), world, u, rank)
        return rc, comm

    def all_reduce_sum_f32(self, t: torch.Tensor, comm, stream: int) -> int:
        return self._lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), self._FLOAT32, self._SUM, comm, stream)

    def async_error(self, comm):
        err = self._C.c_int(0)
        rc = self._lib.ncclCommGetAsyncError(comm, self._C.byref(err))
        return rc, err.value

    def destroy(self, comm) -> int:
        return self._lib.ncclCommDestroy(comm)

    def abort(self, comm) -> int:
        return self._lib.ncclCommAbort(comm)

    def error_string(self, rc: int) -> str:
        return self._lib.ncclGetErrorString(rc).decode()


def _torch_rccl_api(group) -> RcclApi:
    """Default loader: usable only when the process group itself runs on RCCL."""
    if dist.get_backend(group) != "nccl":
        raise RuntimeError("process group backend is %s, not nccl" % dist.get_backend(group))
    return RcclApi()


class DirectRccl:
    """One RCCL communicator of our own, driven through ctypes, so that the gradient all-reduce is enqueued ON THE CALLER'S
    STREAM: ncclAllReduce(bucket, bucket, n, ncclFloat32, ncclSum, comm, current stream) sits between wgnn_finish(6) and
    wgnn_finish(0, adam) like any other kernel of the step.  torch.distributed's NCCL backend runs every collective on a
    stream of its own and brackets it with two event dependencies; on this platform each such hop idles the queue for
    ~10-20 us (profiles/r3_collective_path_one_rank.txt), i.e. more than the 0.67 MB all-reduce itself.

    OPT-IN (WGNN_RCCL_DIRECT=1, `TrainStep(direct_rccl=True)`, `bench.py --direct-rccl`): the communicator lives outside
    torch's ProcessGroupNCCL (no watchdog of its own) and no box with two GPUs has run it yet.

    Bootstrap (the canonical one): rank 0 draws ncclGetUniqueId, the 128 bytes travel through the EXISTING torch process
    group together with a 129th byte -- rank 0's status --, every rank calls ncclCommInitRank.  EVERY decision is collective
    and is taken BEFORE any rank enters a call that could wait for the others (round 4's version let rank 0 skip
    ncclCommInitRank after a failed draw while the other ranks entered it with a zero id):
      1. "I can load the library" -- MIN all-reduce; 0 -> nobody goes on;
      2. rank 0's draw -- its status byte rides with the id; != ok -> nobody calls ncclCommInitRank;
      3. "my ncclCommInitRank returned success" -- MIN all-reduce; 0 -> whoever has a communicator ABORTS it (ncclCommAbort:
         never a call that could wait for a peer that has none);
      4. known-answer all-reduce, [1, rank + 1] -> [world, world (world + 1) / 2]: "my enqueue was accepted" is agreed on by a
         MIN all-reduce BEFORE any rank waits for the result; the wait itself is bounded (event query + ncclCommGetAsyncError
         until `probe_timeout_s`), then "my answer is right" -- MIN all-reduce; 0 -> abort.
    Whenever the answer is no, `comm` stays None on every rank and BucketExchange uses dist.all_reduce.
    `check()` polls ncclCommGetAsyncError (TrainStep calls it every `check_every` steps)."""

    def __init__(self, device: torch.device, group=None, loader=None, probe_timeout_s: float = 30.0):
        self.comm = None
        self.api = None
        self.why = None                        # why the direct path is not in use (None while it is)
        self.timeout_s = probe_timeout_s
        self.device = torch.device(device)
        cuda = self.device.type == "cuda"

        def agree(flag: int) -> bool:          # every rank calls this the same number of times, whatever its own state
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1

        api = None
        try:
            api = (loader or _torch_rccl_api)(group)
        except Exception as e:                 # library missing, wrong backend, symbol missing
            self.why = "library not usable on this rank: %r" % (e,)
        if not agree(api is not None):
            self.why = self.why or "the RCCL library is not usable on some rank"
            return
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        # ---- 2: the id and rank 0's status travel together
        payload = bytearray(129)
        if rank == 0:
            try:
                rc0, uid = api.get_unique_id()
            except Exception:
                rc0, uid = -1, bytes(128)
            payload[:128] = uid[:128]
            payload[128] = 1 if rc0 == 0 else 0
        t = torch.tensor(list(payload), dtype=torch.uint8).to(self.device)
        dist.broadcast(t, src=src, group=group)
        raw = bytes(t.cpu().tolist())
        if raw[128] != 1:
            self.why = "ncclGetUniqueId failed on rank 0"
            return                              # every rank read the same byte: nobody calls ncclCommInitRank
        # ---- 3
        if cuda:
            torch.cuda.set_device(self.device)
        comm = None
        try:
            rc, comm = api.comm_init_rank(world, raw[:128], rank)
        except Exception:
            rc = -1
        if not agree(rc == 0):
            if rc == 0:
                self._abort(api, comm)
            self.why = "ncclCommInitRank failed on some rank"
            return
        # ---- 4: known answer, enqueue agreed on before anybody waits
        self.api, self.comm = api, comm
        probe = torch.tensor([1.0, float(rank + 1), 0.0, 0.0], dtype=torch.float32, device=self.device)
        try:
            self._enqueue(probe)
            enq = True
        except Exception:
            enq = False
        fine = False
        if agree(enq):
            fine = self._wait_probe(probe, probe_timeout_s)
            if fine:
                got = probe.cpu().tolist()
                fine = got[0] == float(world) and got[1] == world * (world + 1) / 2.0
            why = "the known-answer all-reduce returned a wrong sum or did not complete on some rank"
        else:
            why = "ncclAllReduce was refused at enqueue on some rank"
        if not agree(fine):
            self._abort(api, comm)
            self.api, self.comm, self.why = None, None, why

    @staticmethod
    def _abort(api, comm) -> None:
        try:
            api.abort(comm)
        except Exception:
            pass

    def _wait_probe(self, probe: torch.Tensor, timeout_s: float) -> bool:
        """Bounded wait for the probe all-reduce: False on an asynchronous communicator error or after `timeout_s`."""
        if probe.device.type != "cuda":          # a stand-in without streams completes in its own wait(comm, timeout_s)
            return bool(self.api.wait(self.comm, timeout_s)) if hasattr(self.api, "wait") else True
        import time
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(probe.device))
        t0 = time.monotonic()
        while not ev.query():
            rc, err = self.api.async_error(self.comm)
            if rc != 0 or err != 0 or time.monotonic() - t0 > timeout_s:
                return False
            time.sleep(0.001)
        return True

    def _enqueue(self, t: torch.Tensor) -> None:
        stream = torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0
        rc = self.api.all_reduce_sum_f32(t, self.comm, stream)
        if rc != 0:
            raise RuntimeError("windgnn_amd: ncclAllReduce failed: %s" % self.api.error_string(rc))

    def all_reduce_(self, t: torch.Tensor) -> None:
        """In-place fp32 sum over the ranks, enqueued on the current stream of t's device (stream-ordered: no host wait)."""
        self._enqueue(t)
        if not t.is_cuda and hasattr(self.api, "wait") and not self.api.wait(self.comm, self.timeout_s):
            raise RuntimeError("windgnn_amd: the all-reduce did not complete within %.0f s" % self.timeout_s)

    def check(self) -> None:
        """ncclCommGetAsyncError: raise if the communicator reported an asynchronous error since the last check."""
        if self.comm is None:
            return
        rc, err = self.api.async_error(self.comm)
        if rc != 0 or err != 0:
            raise RuntimeError("windgnn_amd: the step's RCCL communicator reports an asynchronous error: %s"
                               % self.api.error_string(err if err != 0 else rc))

    def close(self) -> None:
        if self.comm is not None:
            self.api.destroy(self.comm)
            self.comm = None


class BucketExchange:
    """The collectives of one data-parallel step over a flat bucket `[4-float header | conv grads | GRU grads]`.

    Every collective here is issued unconditionally by every rank in the same order, whatever the rank-local state
    (a cache keyed by the local window count let one rank skip a collective another one issued).
    The global window count either comes from the caller (`n_global`, e.g. bench.py's fixed world * B: no extra
    collective, no host sync) or is all-reduced on EVERY call."""

    def __init__(self, bucket: torch.Tensor, n_conv: int, group=None, direct=None, rccl_loader=None):
        self.bucket = bucket
        self.n_conv = n_conv
        self.group = group
        self.world = dist.get_world_size(group)
        # OPT-IN (WGNN_RCCL_DIRECT=1 or direct=True; every rank must ask for the same -- the bootstrap is collective): the single
        # all-reduce on the compute stream through our own RCCL communicator; the default is torch.distributed's collective on
        # its own stream (+8 us per step at one rank, but under ProcessGroupNCCL's watchdog).  `rccl_loader`: see DirectRccl.
        if direct is None:
            direct = os.environ.get("WGNN_RCCL_DIRECT", "0") == "1"
        self.direct = None
        self.direct_declined = None              # why an asked-for direct path is not in use
        if direct and (bucket.is_cuda or rccl_loader is not None):
            d = DirectRccl(bucket.device, group, rccl_loader)
            if d.comm is not None:
                self.direct = d
            else:
                self.direct_declined = d.why

    def ranks_seen(self) -> int:
        """An all-reduce of ones through the SAME path the step's bucket takes (the own communicator when it is in use, else
        torch.distributed): the number of ranks that really took part.  bench.py puts it in its JSON line, so that a scaling
        record proves N ranks exchanged data."""
        t = torch.ones(4, dtype=torch.float32, device=self.bucket.device)
        if self.direct is not None:
            self.direct.all_reduce_(t)
        else:
            dist.all_reduce(t, group=self.group)
        return int(round(float(t[0].item())))

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
