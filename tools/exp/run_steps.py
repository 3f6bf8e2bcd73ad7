"""a few training steps at the bench shape (debug builds that print from inside the kernels): run_steps.py [n] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m = GCN_GRU(F, F, F, S * F, H, math=os.environ.get("MATH", "f16x3")).to(dev)
tr = TrainStep(m)
A = adjacency_34().to(dev)
X, L = make_inputs(B, 0, dev)
for _ in range(n):
    tr.step(A, X, L)
torch.cuda.synchronize()
