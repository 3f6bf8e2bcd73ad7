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
