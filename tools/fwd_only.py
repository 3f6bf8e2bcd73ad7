"""Forward-only (no stash: inference, wgnn_fwd(stash = NULL)) at the bench workload, for the profiler passes.
    [MATH=f16x3|f16] [IO=fp32|bf16] [WGNN_FUSED_FWD=0|1|2] python tools/fwd_only.py [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU
from windgnn_amd.functional import gcn_gru_forward_raw, prepared_weights
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
m = GCN_GRU(F, F, F, S * F, H, math=os.environ.get("MATH", "f16x3")).to(dev)
tr = TrainStep(m)
A = adjacency_34().to(dev)
X, L = make_inputs(4096, 0, dev, io=os.environ.get("IO", "fp32"))
tr.step(A, X, L)                                   # builds the prepared W_ih images
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    gcn_gru_forward_raw(A, X, tr.p_views, m.math, want_stash=False, prepared=tr._prepared)
torch.cuda.synchronize()
