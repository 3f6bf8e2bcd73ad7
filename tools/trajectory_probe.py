"""Loss trajectories of N optimiser steps at the bench workload (B windows, S=34, T=24, H=102) in the math modes, one
fixed batch: prints per step the loss of exact fp32 and the relative difference of the other modes to it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
B, N = int(os.environ.get("B", "4096")), int(os.environ.get("N", "30"))
A = adjacency_34().to(dev)
X, L = make_inputs(B, 0, dev)
traj, final = {}, {}
for math in ("f32", "f16x3", "f16x3g", "f16"):
    torch.manual_seed(0)
    tr = TrainStep(GCN_GRU(F, F, F, S * F, H, math=math).to(dev), lr=1e-3)
    traj[math] = [float(tr.step(A, X, L)[0]) for _ in range(N)]
    final[math] = tr.flat_p.clone()
for i in range(N):
    r = traj["f32"][i]
    print("step %2d  f32 %.8f   f16x3 %+.2e  f16x3g %+.2e  f16 %+.2e" % (i + 1, r, traj["f16x3"][i] / r - 1, traj["f16x3g"][i] / r - 1, traj["f16"][i] / r - 1))
for math in ("f16x3", "f16x3g", "f16"):
    d = (final[math] - final["f32"]).abs()
    print("params after %d steps vs f32: %-7s max %.3e  mean %.3e  frac>1e-4 %.4f" % (N, math, float(d.max()), float(d.mean()), float((d > 1e-4).float().mean())))
