"""r4: same-process A/B of the data-parallel step's fixed cost with ONE rank (the all-reduce moves nothing):
fused single-rank step | collective step through torch.distributed (its own stream) | through the step's own RCCL
communicator on the compute stream (distributed.DirectRccl).  Blocks of 50 steps, interleaved, 6 rounds, medians."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from windgnn_amd.distributed import ensure_rccl_env
ensure_rccl_env()
import torch, torch.distributed as dist
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
    os.environ.setdefault(k, v)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU
from windgnn_amd.trainer import TrainStep
A = adjacency_34().to(dev)
X, L = make_inputs(4096, 0, dev)
math = os.environ.get("MATH", "f16x3")
def mk(**kw):
    torch.manual_seed(0)
    return TrainStep(GCN_GRU(F, F, F, S * F, H, math=math).to(dev), **kw)
trs = {"fused (no group)": mk(), "collective, torch stream": mk(process_group=dist.group.WORLD, direct_rccl=False),
       "collective, direct rccl": mk(process_group=dist.group.WORLD, direct_rccl=True)}
assert trs["collective, direct rccl"].exchange.direct is not None
for tr in trs.values():
    for _ in range(20):
        tr.step(A, X, L, 4096)
res = {k: [] for k in trs}
for rnd in range(6):
    for k, tr in trs.items():
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            tr.step(A, X, L, 4096)
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) * 1e3 / 50)
base = statistics.median(res["fused (no group)"])
for k, v in res.items():
    print("%-28s median %.1f us/step (%+.1f)   runs: %s" % (k, statistics.median(v), statistics.median(v) - base,
                                                           " ".join("%.1f" % x for x in v)))
for tr in trs.values():
    tr.close()
dist.destroy_process_group()
