"""N > 1 path on CPU (gloo, world_size 2 and 3): sharding + the bucket exchange (two all-reduces per step) reproduce the
big-batch gradient and loss, with unequal shards and shard sizes that change between steps.  The per-rank compute is the ORACLE here (test
infrastructure standing in for the HIP kernels, which need a GPU); what is under test is the host
logic of windgnn_amd/distributed.py that bench.py / TrainStep use on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PARAM_KEYS, load_fixture, max_abs, rel_to_max


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import windgnn_oracle as orc
    from windgnn_amd import distributed as wd
    fx = load_fixture("f2_s7_t12_b32_ckpt")
    A, X, L = (torch.from_numpy(fx[k]) for k in ("A", "X", "L"))
    p = {k: v.clone() for k, v in fx["params"].items()}
    sizes = [p[k].numel() for k in PARAM_KEYS]
    bucket = torch.zeros(wd.HEADER + sum(sizes))
    ex = wd.BucketExchange(bucket, sum(sizes[:4]))             # the class TrainStep and bench.py run
    # Global batches of 31, 32, 29 and 32 windows: with two ranks the shards are (16,15), (16,16), (15,14), (16,16) --
    # rank 0's local count repeats while the global batch changes (a per-rank cache of the shard weight
    # then skipped a collective the other rank issued), shards are unequal, and sizes change between steps.
    for step, n_glob in enumerate((31, 32, 29, 32)):
        Xs, Ls = wd.shard_windows(X[:n_glob], L[:n_glob], rank, world)
        # even steps: the count is all-reduced (every rank, every step); odd steps: the caller states it
        w = ex.shard_weight(Xs.shape[0], None if step % 2 == 0 else n_glob)
        Y, cache = orc.forward(A, Xs, p)
        loss_local, dY = orc.mse_loss_and_grad(Y, Ls)
        grads = orc.backward(A, Xs, p, Y, cache, dY * w)
        bucket[wd.LOSS_SLOT] = loss_local
        bucket[wd.HEADER:] = wd.flatten([grads[k] for k in PARAM_KEYS])
        if step == 0:
            ex.all_reduce_all(w)             # TrainStep's default: ONE all-reduce of [loss | conv | GRU gradients]
        elif step % 2 == 0:
            ex.finish(ex.start_gru(), w)
        else:                                # the two-collective form: the conv all-reduce started async, both joined afterwards
            wg = ex.start_gru()
            wc = ex.start_conv(w)
            wg.wait()
            wc.wait()
        np.save(os.path.join(out_dir, "bucket_%d_rank%d.npy" % (step, rank)), bucket.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    from windgnn_amd.distributed import shard_range
    for n in (1, 7, 32, 4097):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_bucket_exchange_equals_big_batch_with_unequal_and_changing_shards(tmp_path, world):
    from oracle import windgnn_oracle as orc
    from windgnn_amd import distributed as wd
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    fx = load_fixture("f2_s7_t12_b32_ckpt")
    A, X, L = (torch.from_numpy(fx[k]) for k in ("A", "X", "L"))
    for step, n_glob in enumerate((31, 32, 29, 32)):
        _, loss, g = orc.train_step(A, X[:n_glob], L[:n_glob], fx["params"])
        ref = torch.cat([g[k].reshape(-1) for k in PARAM_KEYS])
        if n_glob == 32:                                                   # the reference's own big-batch gradient
            ref = torch.cat([fx["grads"][k].reshape(-1) for k in PARAM_KEYS])
        for rank in range(world):                                          # every rank ends with the same bucket
            b = torch.from_numpy(np.load(os.path.join(str(tmp_path), "bucket_%d_rank%d.npy" % (step, rank))))
            assert rel_to_max(b[wd.HEADER:], ref) <= 2e-5, (step, rank)
            assert abs(float(b[wd.LOSS_SLOT]) - float(loss)) <= 1e-6 * max(1.0, float(loss)), (step, rank)


def _worker_empty(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import windgnn_oracle as orc
    from windgnn_amd import distributed as wd
    fx = load_fixture("f2_s7_t12_b32_ckpt")
    A, X, L = (torch.from_numpy(fx[k]) for k in ("A", "X", "L"))
    p = {k: v.clone() for k, v in fx["params"].items()}
    sizes = [p[k].numel() for k in PARAM_KEYS]
    bucket = torch.zeros(wd.HEADER + sum(sizes))
    ex = wd.BucketExchange(bucket, sum(sizes[:4]))
    for step, n_glob in enumerate((2, 1)):                       # world 3: shards (1,1,0) then (1,0,0)
        Xs, Ls = wd.shard_windows(X[:n_glob], L[:n_glob], rank, world)
        w = ex.shard_weight(Xs.shape[0], None if step == 0 else n_glob)
        bucket.zero_()
        if Xs.shape[0] > 0:
            Y, cache = orc.forward(A, Xs, p)
            loss_local, dY = orc.mse_loss_and_grad(Y, Ls)
            grads = orc.backward(A, Xs, p, Y, cache, dY * w)
            bucket[wd.LOSS_SLOT] = loss_local
            bucket[wd.HEADER:] = wd.flatten([grads[k] for k in PARAM_KEYS])
        else:
            assert w == 0.0                                      # what TrainStep._empty_shard_step does: a zero bucket
        ex.all_reduce_all(w)
        np.save(os.path.join(out_dir, "ebucket_%d_rank%d.npy" % (step, rank)), bucket.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_with_an_empty_shard_still_join_every_collective(tmp_path):
    """Fewer windows than ranks (world 3, global batches of 2 and 1): the empty ranks contribute a zero bucket with weight 0
    and issue the same collectives; every rank ends with the big-batch gradient and loss, nobody hangs."""
    from oracle import windgnn_oracle as orc
    from windgnn_amd import distributed as wd
    mp.spawn(_worker_empty, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    fx = load_fixture("f2_s7_t12_b32_ckpt")
    A, X, L = (torch.from_numpy(fx[k]) for k in ("A", "X", "L"))
    for step, n_glob in enumerate((2, 1)):
        _, loss, g = orc.train_step(A, X[:n_glob], L[:n_glob], fx["params"])
        ref = torch.cat([g[k].reshape(-1) for k in PARAM_KEYS])
        for rank in range(3):
            b = torch.from_numpy(np.load(os.path.join(str(tmp_path), "ebucket_%d_rank%d.npy" % (step, rank))))
            assert rel_to_max(b[wd.HEADER:], ref) <= 2e-5, (step, rank)
            assert abs(float(b[wd.LOSS_SLOT]) - float(loss)) <= 1e-6 * max(1.0, float(loss)), (step, rank)


def test_shard_weight_rejects_an_impossible_global_count():
    """a caller-stated n_global smaller than the local shard is a bug upstream: loud, not a silently wrong scale"""
    import types
    from windgnn_amd import distributed as wd
    ex = types.SimpleNamespace(bucket=torch.zeros(8), group=None)
    with pytest.raises(RuntimeError):
        wd.BucketExchange.shard_weight(ex, 16, 8)
    assert wd.BucketExchange.shard_weight(ex, 16, 32) == 0.5


# ------------------------------------------------------------------------------------------------------------------------
# DirectRccl's bootstrap protocol against a stand-in library.
# The real loader is hard-wired to torch's librccl.so and needs >= 2 GPUs to say anything; what can go wrong in the
# PROTOCOL -- a rank entering ncclCommInitRank while another has decided not to, a rank waiting for a collective a peer never
# enqueued -- is independent of the library, so it is exercised here with a fake whose data path is a directory of files
# (enqueue = write my contribution, asynchronous like a stream-ordered ncclAllReduce; wait = bounded poll for the peers').

class FakeRccl:
    UID = bytes([7] * 128)

    def __init__(self, root, rank, world, mode):
        self.root, self.rank, self.world, self.mode = root, rank, world, mode
        self.seq = 0
        self.pending = None

    def _mark(self, what):
        open(os.path.join(self.root, "%s_rank%d" % (what, self.rank)), "w").close()

    def get_unique_id(self):
        self._mark("draw")
        return (5, bytes(128)) if self.mode == "draw_fails" else (0, self.UID)

    def comm_init_rank(self, world, uid, rank):
        self._mark("init_entered")
        if uid != self.UID or world != self.world or rank != self.rank:
            return 4, None
        if self.mode == "init_fails" and rank == 1:
            return 3, None
        return 0, "comm%d" % rank

    def all_reduce_sum_f32(self, t, comm, stream):
        if self.mode == "enqueue_fails" and self.rank == 1:
            return 2
        self.seq += 1
        if not (self.mode == "peer_silent" and self.rank == 1):      # a peer that accepts the call and never sends
            tmp = os.path.join(self.root, "tmp_%d_%d.npy" % (self.seq, self.rank))
            np.save(tmp, t.numpy())
            os.replace(tmp, os.path.join(self.root, "ar_%d_%d.npy" % (self.seq, self.rank)))
        self.pending = (self.seq, t)
        return 0

    def wait(self, comm, timeout_s):
        import time
        seq, t = self.pending
        paths = [os.path.join(self.root, "ar_%d_%d.npy" % (seq, r)) for r in range(self.world)]
        t0 = time.monotonic()
        while not all(os.path.exists(q) for q in paths):
            if time.monotonic() - t0 > timeout_s:
                return False
            time.sleep(0.002)
        total = sum(torch.from_numpy(np.load(q)) for q in paths)
        if self.mode == "wrong_sum" and self.rank == 0:
            total = total + 1.0
        t.copy_(total)
        return True

    def async_error(self, comm):
        return 0, 0

    def destroy(self, comm):
        self._mark("destroyed")
        return 0

    def abort(self, comm):
        self._mark("aborted")
        return 0

    def error_string(self, rc):
        return "fake rccl error %d" % rc


def _worker_fake_rccl(rank, world, port, out_dir, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from windgnn_amd import distributed as wd

    def loader(group):
        if mode == "load_fails" and rank == world - 1:
            raise OSError("librccl.so: cannot open shared object file")
        return FakeRccl(out_dir, rank, world, mode)

    bucket = torch.zeros(wd.HEADER + 40)
    d = wd.DirectRccl(torch.device("cpu"), None, loader, probe_timeout_s=1.0)     # the protocol itself, with a short probe wait
    ex = wd.BucketExchange(bucket, 8, direct=False)
    ex.direct = d if d.comm is not None else None
    # whatever was decided, the step's collective works and every rank takes part
    bucket[wd.LOSS_SLOT] = float(rank + 1)
    bucket[wd.HEADER:] = torch.arange(40, dtype=torch.float32) * (rank + 1)
    ex.all_reduce_all(1.0)
    seen = ex.ranks_seen()
    np.save(os.path.join(out_dir, "fbucket_rank%d.npy" % rank), bucket.numpy())
    with open(os.path.join(out_dir, "result_rank%d" % rank), "w") as f:
        f.write("%d %d %s" % (1 if d.comm is not None else 0, seen, d.why))
    d.check()
    d.close()
    dist.barrier()
    dist.destroy_process_group()


def _spawn_with_deadline(fn, args, nprocs, deadline_s):
    """mp.spawn in fresh child processes; a rank that blocks (the failure this protocol is about) fails the test instead of
    hanging it."""
    import time
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    t0 = time.monotonic()
    while not ctx.join(timeout=1.0):
        if time.monotonic() - t0 > deadline_s:
            for pr in ctx.processes:
                if pr.is_alive():
                    pr.kill()
            pytest.fail("a rank blocked for more than %d s" % deadline_s)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["ok", "load_fails", "draw_fails", "init_fails", "enqueue_fails", "peer_silent", "wrong_sum"])
def test_direct_rccl_bootstrap_is_all_or_none_and_never_blocks(tmp_path, world, mode):
    """(i) rank 0's id draw fails, (ii) one rank's init fails, (iii) the known-answer all-reduce returns a wrong sum -- and: the
    library does not load on one rank, one rank's enqueue is refused, a peer accepts the enqueue and never sends.  In every
    failing mode EVERY rank must end without the communicator and fall back to dist.all_reduce; none may block."""
    from windgnn_amd import distributed as wd
    out = str(tmp_path)
    _spawn_with_deadline(_worker_fake_rccl, (world, _free_port(), out, mode), world, 60)
    took = []
    for rank in range(world):
        on, seen, why = open(os.path.join(out, "result_rank%d" % rank)).read().split(" ", 2)
        took.append(int(on))
        assert int(seen) == world, (rank, seen)
        b = torch.from_numpy(np.load(os.path.join(out, "fbucket_rank%d.npy" % rank)))
        tri = world * (world + 1) / 2.0
        assert float(b[wd.LOSS_SLOT]) == tri and torch.equal(b[wd.HEADER:], torch.arange(40, dtype=torch.float32) * tri), rank
    assert took == [1 if mode == "ok" else 0] * world, (mode, took)
    have = lambda what, r: os.path.exists(os.path.join(out, "%s_rank%d" % (what, r)))
    if mode in ("load_fails", "draw_fails"):          # nobody may enter ncclCommInitRank (round 4: ranks != 0 did, with a zero id)
        assert not any(have("init_entered", r) for r in range(world))
    if mode == "init_fails":                           # ranks that got a communicator abort it; rank 1 has none to release
        assert [have("aborted", r) for r in range(world)] == [r != 1 for r in range(world)]
    if mode in ("enqueue_fails", "peer_silent", "wrong_sum"):
        assert all(have("aborted", r) for r in range(world))
    if mode == "ok":
        assert all(have("destroyed", r) and not have("aborted", r) for r in range(world))
