"""Forward-only (no stash) timing of the one-pass fp16 / bf16-I/O configuration, per kernel (library event profiler).
    python tools/exp/fwd_only_f16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.functional import gcn_gru_forward_raw
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = GCN_GRU(F, F, F, S * F, H, math=os.environ.get("MATH", "f16")).to(dev)
tr = TrainStep(m)
A = adjacency_34().to(dev)
X, L = make_inputs(4096, 0, dev, io=os.environ.get("IO", "bf16"))
tr.step(A, X, L)
fwd = lambda: gcn_gru_forward_raw(A, X, tr.p_views, m.math, want_stash=False, prepared=tr._prepared)
for _ in range(20):
    fwd()
torch.cuda.synchronize()
_lib.profile_enable(True)
n = 50
for _ in range(n):
    fwd()
torch.cuda.synchronize()
recs = _lib.profile_read()
_lib.profile_enable(False)
print("  ".join("%s=%.1f" % (r["name"], 1e3 * r["ms"] / r["launches"]) for r in recs))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    fwd()
e1.record()
torch.cuda.synchronize()
print("forward = %.1f us" % (e0.elapsed_time(e1) * 1e3 / 100))
