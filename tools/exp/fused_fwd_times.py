"""Forward-only kernel times, fused front end (gcngi) vs the two launches it replaces.  MATH=f16x3|f16 IO=fp32|bf16"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.functional import gcn_gru_forward_raw
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
math, io = os.environ.get("MATH", "f16x3"), os.environ.get("IO", "fp32")
m = GCN_GRU(F, F, F, S * F, H, math=math).to(dev)
tr = TrainStep(m)
A = adjacency_34().to(dev)
X, L = make_inputs(4096, 0, dev, io=io)
tr.step(A, X, L)
for fused in (0, 2):
    _lib.set_option(_lib.OPT_FUSED_FWD, fused)     # (the environment variable is read once; round 5)
    for what in ("forward only (no stash)", "training step"):
        fn = (lambda: gcn_gru_forward_raw(A, X, tr.p_views, m.math, want_stash=False, prepared=tr._prepared)) \
            if what.startswith("forward") else (lambda: tr.step(A, X, L))
        for _ in range(10):
            fn()
        _lib.profile_enable(True)
        n = 20
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        recs = _lib.profile_read()
        _lib.profile_enable(False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("fused=%s %s %s io=%s: %.1f us   [%s]" % (fused, math, what, io, e0.elapsed_time(e1) * 1e3 / 50,
              "  ".join("%s=%.1f" % (r["name"], 1e3 * r["ms"] / r["launches"]) for r in recs)), flush=True)
