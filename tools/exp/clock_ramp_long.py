"""Strict training step (B = 4096): averages over consecutive 20-step windows for 600 steps after a host synchronisation + 20 ms of
idling -- how long the GPU's clocks take to settle, and how steady the plateau is on this box.    python tools/exp/clock_ramp_long.py"""
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
for trial in range(4):
    torch.cuda.synchronize()
    time.sleep(0.02)
    nwin = 30
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nwin + 1)]
    ev[0].record()
    for w in range(nwin):
        for _ in range(20):
            tr.step(A, X, L)
        ev[w + 1].record()
    torch.cuda.synchronize()
    print("trial %d, us per step in windows of 20 steps: %s" % (trial, " ".join("%.0f" % (ev[w].elapsed_time(ev[w + 1]) * 50) for w in range(nwin))), flush=True)
