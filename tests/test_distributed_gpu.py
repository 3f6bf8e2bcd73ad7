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


def _train(rank, world, port, out_dir, backend="gloo", tag=None, batches=(32, 32), stated=False, overlap=False, direct=None):
    """`batches`: the global batch of each step (sharded over the ranks); `stated`: pass n_global to TrainStep.step
    instead of letting the exchange all-reduce the count; `overlap`: the two-collective form (the GRU gradients' all-reduce
    overlapped with backward part 2) instead of the default single all-reduce."""
    from windgnn_amd.distributed import ensure_rccl_env, shard_windows
    ensure_rccl_env()                     # the environment bench.py establishes, before this process touches the GPU
    from windgnn_amd import GCN_GRU
    from windgnn_amd.trainer import TrainStep
    group = None
    if world > 1 or backend == "nccl":
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        if backend == "nccl":
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
            group = dist.group.WORLD            # an explicit group: TrainStep runs its collective path even with 1 rank
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    fx = load_fixture("f2_s7_t12_b32_ckpt")
    m = GCN_GRU(13, 13, 13, 7 * 13, 21, math="f16x3")
    m.load_state_dict({k: v.clone() for k, v in fx["params"].items()})
    m = m.to(dev)
    tr = TrainStep(m, process_group=group, overlap_collectives=overlap, direct_rccl=direct)
    assert tr.collective == (world > 1 or backend == "nccl")
    if backend == "nccl" and not overlap and direct is True:
        assert tr.exchange.direct is not None, tr.exchange.direct_declined   # our own RCCL communicator (opt-in), on the compute stream
    elif tr.collective:
        assert tr.exchange.direct is None                # the default: torch.distributed's collective
    A = torch.from_numpy(fx["A"]).to(dev)
    X, L = torch.from_numpy(fx["X"]), torch.from_numpy(fx["L"])
    losses = []
    for n_glob in batches:
        Xs, Ls = shard_windows(X[:n_glob], L[:n_glob], rank, world)
        loss, _ = tr.step(A, Xs.to(dev), Ls.to(dev), n_global=n_glob if stated else None)
        losses.append(float(loss))
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(out_dir, "p_%s.npy" % (tag or "world%d" % world)), tr.flat_p.cpu().numpy())
        np.save(os.path.join(out_dir, "loss_%s.npy" % (tag or "world%d" % world)), np.array(losses))
    if world > 1 or backend == "nccl":
        dist.barrier()
        tr.close()
        dist.destroy_process_group()


def test_two_rank_training_equals_single_process(tmp_path):
    assert torch.cuda.is_available()
    _train(0, 1, 0, str(tmp_path))
    mp.spawn(_train, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_train, args=(2, _free_port(), str(tmp_path), "gloo", "world2_overlap", (32, 32), False, True), nprocs=2, join=True)
    p1 = torch.from_numpy(np.load(os.path.join(str(tmp_path), "p_world1.npy")))
    l1 = np.load(os.path.join(str(tmp_path), "loss_world1.npy"))
    for tag in ("world2", "world2_overlap"):      # the single all-reduce (default) and the two-collective form
        p2 = torch.from_numpy(np.load(os.path.join(str(tmp_path), "p_%s.npy" % tag)))
        # Adam moves each weight by ~1e-3 per step; shard-sum vs big-batch gradients differ only by fp32 rounding
        assert max_abs(p1, p2) <= 2e-5, tag
        # the loss every rank returns is the big-batch mean loss, not its shard's
        l2 = np.load(os.path.join(str(tmp_path), "loss_%s.npy" % tag))
        assert np.abs(l1 - l2).max() <= 1e-6 * max(1.0, float(np.abs(l1).max())), tag


def test_two_rank_training_with_unequal_and_changing_shards(tmp_path):
    """Global batches of 31, 32, 29, 32 windows over two ranks: shards (16,15), (16,16), (15,14), (16,16).  Rank 0's
    local count repeats while the global batch changes (the per-rank shard-weight cache then skipped a
    collective the other rank issued -> mismatched all-reduces); every collective is now issued by every rank on
    every step.  Parameters and losses must equal the single-process run over the same global batches, both when
    the exchange all-reduces the window count and when the caller states it."""
    assert torch.cuda.is_available()
    sched = (31, 32, 29, 32)
    _train(0, 1, 0, str(tmp_path), "gloo", "ref", sched)
    mp.spawn(_train, args=(2, _free_port(), str(tmp_path), "gloo", "counted", sched, False), nprocs=2, join=True)
    mp.spawn(_train, args=(2, _free_port(), str(tmp_path), "gloo", "stated", sched, True), nprocs=2, join=True)
    p1 = torch.from_numpy(np.load(os.path.join(str(tmp_path), "p_ref.npy")))
    l1 = np.load(os.path.join(str(tmp_path), "loss_ref.npy"))
    for tag in ("counted", "stated"):
        p2 = torch.from_numpy(np.load(os.path.join(str(tmp_path), "p_%s.npy" % tag)))
        assert max_abs(p1, p2) <= 4e-5, tag          # four Adam steps
        l2 = np.load(os.path.join(str(tmp_path), "loss_%s.npy" % tag))
        assert np.abs(l1 - l2).max() <= 1e-6 * max(1.0, float(np.abs(l1).max())), tag


def test_two_rank_training_with_an_empty_shard(tmp_path):
    """Global batches of 1, 32 and 1 windows over two ranks: in steps 1 and 3 rank 1 has NO windows (it used to fail
    its shape check after the count collective and leave rank 0 waiting in the gradient all-reduce).  An empty shard now
    contributes a zero bucket, issues every collective and runs the optimiser launch; parameters and losses must equal the
    single-process run over the same global batches, in both collective forms."""
    assert torch.cuda.is_available()
    sched = (1, 32, 1)
    _train(0, 1, 0, str(tmp_path), "gloo", "ref", sched)
    mp.spawn(_train, args=(2, _free_port(), str(tmp_path), "gloo", "counted", sched, False), nprocs=2, join=True)
    mp.spawn(_train, args=(2, _free_port(), str(tmp_path), "gloo", "stated_overlap", sched, True, True), nprocs=2, join=True)
    p1 = torch.from_numpy(np.load(os.path.join(str(tmp_path), "p_ref.npy")))
    l1 = np.load(os.path.join(str(tmp_path), "loss_ref.npy"))
    for tag in ("counted", "stated_overlap"):
        p2 = torch.from_numpy(np.load(os.path.join(str(tmp_path), "p_%s.npy" % tag)))
        assert max_abs(p1, p2) <= 3e-5, tag
        l2 = np.load(os.path.join(str(tmp_path), "loss_%s.npy" % tag))
        assert np.abs(l1 - l2).max() <= 1e-6 * max(1.0, float(np.abs(l1).max())), tag


def _bench_shape_inputs(n):
    """Seeded inputs of the bench workload's shape (S=34, T=24, H=102), identical in every process."""
    from oracle import windgnn_oracle as orc
    from conftest import GOLDEN
    S, T, H = 34, 24, 102
    A = torch.from_numpy(np.load(os.path.join(GOLDEN, "graph_7_34.npz"))["A34"]).float()
    g = torch.Generator().manual_seed(512)
    X = torch.rand(n, T, S, 13, generator=g)
    L = torch.rand(n, T, H, generator=g)
    return A, X, L, orc.init_params(S, 13, H, seed=0)


def _train_bench_shape(rank, world, port, out_dir, math, splits, tag, stated):
    """bench.py's own step (TrainStep.step at S=34, H=102, >= 4096 rows per rank, so f16x3g's single-plane gate gradients, the
    per-rank power-of-two range scale from an n_local / n_global-prescaled dY, the deferred partials -> wgnn_finish(6) ->
    all-reduce -> wgnn_finish(0, adam) are all active) on this rank's windows [splits[rank], splits[rank + 1])."""
    from windgnn_amd.distributed import ensure_rccl_env
    ensure_rccl_env()
    from windgnn_amd import GCN_GRU
    from windgnn_amd.trainer import TrainStep
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    n = splits[-1]
    A, X, L, p = _bench_shape_inputs(n)
    m = GCN_GRU(13, 13, 13, 34 * 13, 102, math=math)
    m.load_state_dict({k: v.clone() for k, v in p.items()})
    tr = TrainStep(m.to(dev))
    assert tr.collective == (world > 1)
    lo, hi = splits[rank], splits[rank + 1]
    Xs, Ls = X[lo:hi].contiguous().to(dev), L[lo:hi].contiguous().to(dev)
    losses, g1 = [], None
    for step in range(2):
        loss, _ = tr.step(A.to(dev), Xs, Ls, n_global=n if stated else None)
        losses.append(float(loss))
        if step == 0:
            g1 = tr.flat_g.cpu().numpy().copy()          # after the all-reduce: the big-batch gradient on every rank
    tr.check()
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(out_dir, "p_%s.npy" % tag), tr.flat_p.cpu().numpy())
        np.save(os.path.join(out_dir, "g_%s.npy" % tag), g1)
        np.save(os.path.join(out_dir, "loss_%s.npy" % tag), np.array(losses))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("math", ["f16x3g", "f32", "f16x3"])
def test_two_rank_training_at_the_bench_shape_equals_single_process_and_the_oracle(tmp_path, math):
    """The code bench.py --gpus N runs, sharded: S=34, H=102, a global batch of 512 windows over two ranks
    as 256 / 256 and as 257 / 255 (>= 4096 rows per rank), counted and stated n_global.  After step 1 the all-reduced bucket
    must equal the single-process big-batch gradient (and the fp64 oracle's, at SURVEY 8c's bar); both steps' losses and the
    parameters after two Adam steps must equal the single-process run's."""
    from oracle import windgnn_oracle as orc
    from conftest import rel_to_max
    assert torch.cuda.is_available()
    out = str(tmp_path)
    n = 512
    _train_bench_shape(0, 1, 0, out, math, (0, n), "ref", True)
    mp.spawn(_train_bench_shape, args=(2, _free_port(), out, math, (0, 256, n), "even", True), nprocs=2, join=True)
    mp.spawn(_train_bench_shape, args=(2, _free_port(), out, math, (0, 257, n), "uneven", False), nprocs=2, join=True)
    A, X, L, p = _bench_shape_inputs(n)
    _, loss_o, go = orc.train_step(A.double(), X.double(), L.double(), {k: v.double() for k, v in p.items()})
    g_o = torch.cat([go[k].reshape(-1) for k in PARAM_KEYS])
    sizes = [go[k].numel() for k in PARAM_KEYS]
    ref = {k: np.load(os.path.join(out, "%s_ref.npy" % k)) for k in ("p", "g", "loss")}
    assert abs(ref["loss"][0] - float(loss_o)) <= 1e-5 * max(1.0, float(loss_o))
    for tag in ("even", "uneven"):
        got = {k: np.load(os.path.join(out, "%s_%s.npy" % (k, tag))) for k in ("p", "g", "loss")}
        assert np.abs(got["loss"] - ref["loss"]).max() <= 2e-6 * max(1.0, float(np.abs(ref["loss"]).max())), (tag, got["loss"], ref["loss"])
        worst_ref, worst_orc = 0.0, 0.0
        for a, b, c in zip(torch.from_numpy(got["g"]).split(sizes), torch.from_numpy(ref["g"]).split(sizes), g_o.split(sizes)):
            worst_ref = max(worst_ref, rel_to_max(a, b))
            worst_orc = max(worst_orc, rel_to_max(a, c))
        dp = np.abs(got["p"] - ref["p"])
        print("bench-shape 2 ranks %s %s: grad vs single %.1e, vs fp64 oracle %.1e; params max %.1e, beyond 2e-5: %.2e"
              % (math, tag, worst_ref, worst_orc, dp.max(), float((dp > 2e-5).mean())))
        assert worst_ref <= 2e-5, (tag, worst_ref)              # shard sums vs the big batch: rounding only
        assert worst_orc <= 1e-4, (tag, worst_orc)
        # two Adam steps of ~1e-3 each; observed (r4): every parameter within 1.3e-7 of the single-process run's
        assert dp.max() <= 5e-6, (tag, dp.max())


def test_rccl_allreduce_path_executes_and_is_bitwise_neutral_with_one_rank(tmp_path):
    """The RCCL ("nccl") code paths of TrainStep.step -- init_process_group("nccl"), the single all-reduce of the bucket
    (default) and the two-collective form (async all-reduce of the GRU gradients overlapped with backward part 2, then the
    conv-gradient + loss all-reduce under the GRU tensors' optimiser step) -- executed for real in a fresh
    child process on the one GPU this box has (world_size 1: RCCL refuses two ranks on one device).  A one-rank sum
    is the identity, so the parameters after two steps must equal the no-group run bit for bit."""
    assert torch.cuda.is_available()
    mp.spawn(_train, args=(1, _free_port(), str(tmp_path), "gloo", "plain"), nprocs=1, join=True)
    mp.spawn(_train, args=(1, _free_port(), str(tmp_path), "nccl", "rccl1", (32, 32), False, False, True), nprocs=1, join=True)
    mp.spawn(_train, args=(1, _free_port(), str(tmp_path), "nccl", "rccl1_torch"), nprocs=1, join=True)
    mp.spawn(_train, args=(1, _free_port(), str(tmp_path), "nccl", "rccl1_overlap", (32, 32), False, True), nprocs=1, join=True)
    p0 = np.load(os.path.join(str(tmp_path), "p_plain.npy"))
    # the single all-reduce through our own communicator on the compute stream (opt-in: distributed.DirectRccl), the same
    # through torch.distributed (its own stream: the default), and the two-collective form
    for tag in ("rccl1", "rccl1_torch", "rccl1_overlap"):
        assert np.array_equal(p0, np.load(os.path.join(str(tmp_path), "p_%s.npy" % tag))), tag
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "loss_plain.npy")),
                              np.load(os.path.join(str(tmp_path), "loss_%s.npy" % tag))), tag


def _train_nccl_multi(rank, world, port, out_dir, direct):
    """One rank per GPU on the nccl backend: TrainStep with / without the own communicator, bucket and parameters saved."""
    from windgnn_amd.distributed import ensure_rccl_env, shard_windows
    ensure_rccl_env()
    from windgnn_amd import GCN_GRU
    from windgnn_amd.trainer import TrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    fx = load_fixture("f2_s7_t12_b32_ckpt")
    m = GCN_GRU(13, 13, 13, 7 * 13, 21, math="f16x3")
    m.load_state_dict({k: v.clone() for k, v in fx["params"].items()})
    with TrainStep(m.to(dev), process_group=dist.group.WORLD, direct_rccl=direct) as tr:
        assert (tr.exchange.direct is not None) == bool(direct), tr.exchange.direct_declined
        assert tr.exchange.ranks_seen() == world
        A = torch.from_numpy(fx["A"]).to(dev)
        X, L = torch.from_numpy(fx["X"]), torch.from_numpy(fx["L"])
        for n_glob in (31, 32):
            Xs, Ls = shard_windows(X[:n_glob], L[:n_glob], rank, world)
            tr.step(A, Xs.to(dev), Ls.to(dev), n_global=n_glob)
        tr.check()                                   # ncclCommGetAsyncError of the own communicator
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "mg_%d_rank%d.npy" % (int(bool(direct)), rank)),
                np.concatenate([tr._gbuf.cpu().numpy(), tr.flat_p.cpu().numpy()]))
        dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device (the driver's "
                                                          "round-end box has one; this is for an 8-GPU node)")
def test_direct_rccl_with_two_ranks_equals_torch_distributed(tmp_path):
    """the step's own communicator (DirectRccl, opt-in) has only ever moved bytes in a one-rank group.  With two
    devices: two ranks, two steps with unequal shards, through the own communicator and through torch.distributed -- the
    all-reduced bucket and the parameters must agree on every rank and between the two paths (a two-rank fp32 sum is
    order-independent: bit for bit)."""
    for direct in (False, True):
        mp.spawn(_train_nccl_multi, args=(2, _free_port(), str(tmp_path), direct), nprocs=2, join=True)
    a = [np.load(os.path.join(str(tmp_path), "mg_0_rank%d.npy" % r)) for r in range(2)]
    b = [np.load(os.path.join(str(tmp_path), "mg_1_rank%d.npy" % r)) for r in range(2)]
    assert np.array_equal(a[0], a[1]) and np.array_equal(b[0], b[1])
    assert np.array_equal(a[0], b[0])


def test_bench_multi_gpu_code_path_runs_under_torchrun_with_rccl():
    """bench.py's N > 1 path -- RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run, init_process_group("nccl"),
    barriers, the collective TrainStep, max-over-ranks timing, destroy -- rehearsed with ONE rank on the one GPU of this
    box (`--force-dist`); the JSON line must carry the contract's keys and a sane value."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5",
           "--warmup", "2", "--force-dist", "--no-cpu-baseline", "--no-traffic", "--no-secondary", "--batch", "512"]
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}   # bench.py must set it itself
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["value"] > 1e4 and out["scaling"] == "weak"
    assert "rccl" in out["config"]["collective"]
    assert out["rccl_ranks_seen"] == 1               # an all-reduce of ones through the path the step's bucket takes
