"""Host-side cost of one training step (how long the CPU needs to enqueue it) vs its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU
from windgnn_amd.trainer import TrainStep

dev = torch.device("cuda:0")
m = GCN_GRU(F, F, F, S * F, H, math="f16x3").to(dev)
tr = TrainStep(m)
A = adjacency_34().to(dev)
X, L = make_inputs(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 0, dev)
for _ in range(5):
    tr.step(A, X, L)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    tr.step(A, X, L)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
