"""Per-step time of the strict training step (B = 4096) right after a host synchronisation: 40 event-bracketed steps after `pre`
untimed steps, a torch.cuda.synchronize() and an idle gap of `idle` ms (profiles/r5_b_clock_ramp.txt).
    python tools/exp/clock_ramp.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
A = adjacency_34().to(dev)
m = GCN_GRU(F, F, F, S * F, H, math="f16x3").to(dev)
tr = TrainStep(m)
X, L = make_inputs(4096, 0, dev)
for pre, idle_ms in [(5, 0), (5, 0), (50, 0), (5, 5), (5, 50), (200, 0), (5, 0)]:
    for _ in range(pre):
        tr.step(A, X, L)
    torch.cuda.synchronize()
    if idle_ms:
        time.sleep(idle_ms / 1e3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for i in range(40):
        tr.step(A, X, L)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(40)]
    print("pre=%3d idle=%2d ms: first 20 avg %.1f us, last 20 avg %.1f | %s" % (pre, idle_ms, sum(ts[:20]) / 20, sum(ts[20:]) / 20, " ".join("%.0f" % t for t in ts[:24])), flush=True)
