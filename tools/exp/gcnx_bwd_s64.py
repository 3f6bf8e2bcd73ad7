"""r4: gcnx_bwd / gcnx_fwd instance timing at S = 64 (NT = 4) and the one-pass fp32-dg instance (f16 math + wide GRU).
    python tools/exp/gcnx_bwd_s64.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
for S, H, math, B, io in ((64, 102, "f16x3", 2048, torch.float32), (64, 102, "f16", 2048, torch.float32),
                          (64, 102, "f16", 2048, torch.bfloat16),     # r5: the instance that spilled at 12 waves (now 8)
                          (34, 200, "f16", 2048, torch.float32), (34, 102, "f16x3", 4096, torch.float32)):
    T = 24
    torch.manual_seed(0)
    m = GCN_GRU(13, 13, 13, S * 13, H, math=math).to(dev)
    tr = TrainStep(m)
    g = torch.Generator().manual_seed(1)
    A = (torch.rand(S, S, generator=g) / S + 0.01).to(dev)
    X = torch.rand(B, T, S, 13, generator=g).to(io).to(dev)
    L = torch.rand(B, T, H, generator=g).to(io).to(dev)
    for _ in range(5):
        tr.step(A, X, L)
    _lib.profile_enable(True)
    n = 10
    for _ in range(n):
        tr.step(A, X, L)
    torch.cuda.synchronize()
    recs = _lib.profile_read()
    _lib.profile_enable(False)
    print("S=%d H=%d %s %s B=%d: %s" % (S, H, math, str(io).replace("torch.", ""), B, "  ".join("%s=%.1f" % (r["name"], 1e3 * r["ms"] / r["launches"])
                                                                for r in recs if r["name"].startswith("gcnx"))), flush=True)
    del tr, m
