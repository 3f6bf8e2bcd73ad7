"""Per-kernel event timings of the training step at the bench workload (prints a compact table).
    [B=4096] [MATH=f16x3|f32|f16] [IO=fp32|bf16] python tools/kernel_times.py [name filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
m = GCN_GRU(F, F, F, S * F, H, math=os.environ.get("MATH", "f16x3")).to(dev)
tr = TrainStep(m)
A = adjacency_34().to(dev)
X, L = make_inputs(int(os.environ.get("B", "4096")), 0, dev, io=os.environ.get("IO", "fp32"))
for _ in range(20):
    tr.step(A, X, L)
_lib.profile_enable(True)
n = 20
for _ in range(n):
    tr.step(A, X, L)
torch.cuda.synchronize()
recs = _lib.profile_read()
_lib.profile_enable(False)
filt = sys.argv[1] if len(sys.argv) > 1 else ""
print("  ".join("%s=%.1f" % (r["name"], 1e3 * r["ms"] / r["launches"]) for r in recs if filt in r["name"]))
print("sum per step (events) = %.1f us" % sum(1e3 * r["ms"] / n for r in recs))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    tr.step(A, X, L)
e1.record()
torch.cuda.synchronize()
print("step = %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
