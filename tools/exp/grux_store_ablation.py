"""grux_fwd at the bench shape with fewer and fewer stores in its step loop: training forward (gate records + Y + h planes),
stash-less forward (Y only), wgnn_fwd_last (no store inside the loop).  If the step were bound by the bytes it moves the
three would scale with them; if each step waits for its own stores to be acknowledged the last one drops to the chain time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.data import forward_last
from windgnn_amd.functional import gcn_gru_forward_raw
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
Bs = [int(b) for b in sys.argv[1:]] or [4096]
for B in Bs:
    m = GCN_GRU(F, F, F, S * F, H, math=os.environ.get("MATH", "f16x3")).to(dev)
    tr = TrainStep(m)
    A = adjacency_34().to(dev)
    X, L = make_inputs(B, 0, dev)
    tr.step(A, X, L)

    def run(name, fn, n=20):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        recs = _lib.profile_read()
        _lib.profile_enable(False)
        print("B=%d %-12s" % (B, name), "  ".join("%s=%.1f" % (r["name"][:24], 1e3 * r["ms"] / r["launches"]) for r in recs))

    run("train-fwd", lambda: gcn_gru_forward_raw(A, X, tr.p_views, m.math, want_stash=True, labels=L, prepared=tr._prepared))
    run("stash-less", lambda: gcn_gru_forward_raw(A, X, tr.p_views, m.math, want_stash=False, prepared=tr._prepared))
    run("last-only", lambda: forward_last(m, A, X, 0.0, 1.0))
    run("step", lambda: tr.step(A, X, L))
