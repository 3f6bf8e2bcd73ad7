"""The N > 1 trainer path on the GPU box: 2 ranks share cuda:0 (gloo moves the bucket; RCCL would refuse two
ranks on one device), each runs TrainStep.step on its shard; after 2 steps the parameters must equal a
single-process run on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PARAM_KEYS, load_fixture, max_abs

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train(rank, world, port, out_dir):
    from windgnn_amd import GCN_GRU
    from windgnn_amd.distributed import shard_windows
    from windgnn_amd.trainer import TrainStep
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    fx = load_fixture("f2_s7_t12_b32_ckpt")
    m = GCN_GRU(13, 13, 13, 7 * 13, 21, math="f16x3")
    m.load_state_dict({k: v.clone() for k, v in fx["params"].items()})
    m = m.to(dev)
    tr = TrainStep(m)
    A = torch.from_numpy(fx["A"]).to(dev)
    X, L = torch.from_numpy(fx["X"]), torch.from_numpy(fx["L"])
    Xs, Ls = shard_windows(X, L, rank, world)
    Xs, Ls = Xs.contiguous().to(dev), Ls.contiguous().to(dev)
    for _ in range(2):
        tr.step(A, Xs, Ls)
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(out_dir, "p_world%d.npy" % world), tr.flat_p.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_training_equals_single_process(tmp_path):
    assert torch.cuda.is_available()
    _train(0, 1, 0, str(tmp_path))
    mp.spawn(_train, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    p1 = torch.from_numpy(np.load(os.path.join(str(tmp_path), "p_world1.npy")))
    p2 = torch.from_numpy(np.load(os.path.join(str(tmp_path), "p_world2.npy")))
    # Adam moves each weight by ~1e-3 per step; shard-sum vs big-batch gradients differ only by fp32 rounding
    assert max_abs(p1, p2) <= 2e-5
