"""Is the small-batch step (BASELINE configs[1]: B = 256, exact fp32) bound by the GPU or by the host's launch path?
Prints the event-bracketed kernel sum, the GPU step time and the HOST time per step (calls issued without waiting)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
for B, math in ((256, "f32"), (256, "f16x3"), (4096, "f16x3")):
    m = GCN_GRU(F, F, F, S * F, H, math=math).to(dev)
    tr = TrainStep(m)
    A = adjacency_34().to(dev)
    X, L = make_inputs(B, 0, dev)
    for _ in range(30):
        tr.step(A, X, L)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(20):
        tr.step(A, X, L)
    torch.cuda.synchronize()
    recs = _lib.profile_read()
    _lib.profile_enable(False)
    ksum = sum(1e3 * r["ms"] / 20 for r in recs)
    n = 200
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(A, X, L)
    t_host = (time.perf_counter() - t0) / n * 1e6          # the queue absorbs the launches: host time per step
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e6
    print("B=%d %s: kernels (events) %.1f us, step %.1f us, host issues a step in %.1f us, %d launches" %
          (B, math, ksum, t_all, t_host, sum(r["launches"] for r in recs) // 20))
