"""Per-kernel event timings of the forward+backward at the bench workload (prints a compact table)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
m = GCN_GRU(F, F, F, S * F, H, math="f16x3").to(dev)
tr = TrainStep(m)
A = adjacency_34().to(dev)
X, L = make_inputs(int(os.environ.get("B", "4096")), 0, dev)
for _ in range(3):
    tr.forward_backward(A, X, L)
_lib.profile_enable(True)
n = 10
for _ in range(n):
    tr.forward_backward(A, X, L)
torch.cuda.synchronize()
recs = _lib.profile_read()
_lib.profile_enable(False)
filt = sys.argv[1] if len(sys.argv) > 1 else ""
print("  ".join("%s=%.1f" % (r["name"], 1e3 * r["ms"] / r["launches"]) for r in recs if filt in r["name"]))
