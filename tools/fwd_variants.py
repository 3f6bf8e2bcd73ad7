"""Per-kernel times of the forward in its three variants (experiments aid, GPU box):
inference (no stash), training forward (stash), training forward + fused MSE statistics (wgnn_fwd_loss)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import adjacency_34, make_inputs
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.functional import gcn_gru_forward_raw

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
math = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
torch.manual_seed(0)
m = GCN_GRU(13, 13, 13, 34 * 13, 102, math=math).to(dev)
P = [p.detach() for p in m.hot_path_parameters()]
A = adjacency_34().to(dev)
X, L = make_inputs(B, 0, dev)
for name, kw in (("inference", dict(want_stash=False)), ("stash", dict(want_stash=True)),
                 ("stash+loss", dict(want_stash=True, labels=L))):
    for _ in range(20):
        gcn_gru_forward_raw(A, X, P, m.math, **kw)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(50):
        gcn_gru_forward_raw(A, X, P, m.math, **kw)
    torch.cuda.synchronize()
    recs = _lib.profile_read()
    _lib.profile_enable(False)
    print(name, " ".join("%s=%.1fus" % (r["name"], 1e3 * r["ms"] / r["launches"]) for r in sorted(recs, key=lambda r: -r["ms"])[:4]))
