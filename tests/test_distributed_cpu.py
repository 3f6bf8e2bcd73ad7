"""N > 1 path on CPU (gloo, world_size 2): sharding + the single flat-bucket all-reduce reproduce the
big-batch gradient and the reference's Adam step.  The per-rank compute is the ORACLE here (test
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
    Xs, Ls = wd.shard_windows(X, L, rank, world)
    Y, cache = orc.forward(A, Xs, p)
    loss_local, dY = orc.mse_loss_and_grad(Y, Ls)
    dY = dY * wd.grad_scale_for_shard(Xs.shape[0], X.shape[0])
    grads = orc.backward(A, Xs, p, Y, cache, dY)
    flat = wd.flatten([grads[k] for k in PARAM_KEYS])
    wd.allreduce_flat_(flat)
    if rank == 0:
        np.save(os.path.join(out_dir, "flat.npy"), flat.numpy())
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
def test_two_rank_allreduce_equals_big_batch(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    flat = torch.from_numpy(np.load(os.path.join(str(tmp_path), "flat.npy")))
    fx = load_fixture("f2_s7_t12_b32_ckpt")
    ref = torch.cat([fx["grads"][k].reshape(-1) for k in PARAM_KEYS])      # reference big-batch gradient
    assert rel_to_max(flat, ref) <= 2e-5
