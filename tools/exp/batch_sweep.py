"""Training step (forward + MSE + backward + Adam through TrainStep) over the batch size on ONE GPU: windows/s and us per step,
per math mode -- what a data-parallel rank sees when the global batch is cut into more or fewer shards (weak vs strong scaling).
    python tools/exp/batch_sweep.py [math ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
A = adjacency_34().to(dev)
for math in (sys.argv[1:] or ["f16x3", "f16x3g", "f32"]):
    for B in (64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
        m = GCN_GRU(F, F, F, S * F, H, math=math).to(dev)
        tr = TrainStep(m)
        X, L = make_inputs(B, 0, dev)
        n = 100 if B <= 4096 else 30
        best = None
        for rep in range(3):                                     # a host hiccup inside one timed window shows as an outlier: take the best of 3
            for _ in range(10):
                tr.step(A, X, L)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                tr.step(A, X, L)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            best = us if best is None else min(best, us)
        tr.check()
        us = best
        print("%-7s B=%6d  %9.1f us/step  %7.3f M windows/s  (%5.1f ns per window)" % (math, B, us, B / us, 1e3 * us / B), flush=True)
        del tr, m, X, L
        torch.cuda.empty_cache()
