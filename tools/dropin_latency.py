"""The reference's own call shape (src/main.py:64-80: one window of T = 168 steps per step, B = 1) through the
drop-in nn.Module + torch.optim.Adam, and through TrainStep: wall time per training step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import adjacency_34
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
S, T, H = 34, 168, 102
A = adjacency_34().to(dev)
X = torch.rand(1, T, S, 13, device=dev)
L = torch.rand(1, T, H, device=dev)
for math in ("f32", "f16x3"):
    m = GCN_GRU(13, 13, 13, S * 13, H, math=math).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    def ref_style():
        opt.zero_grad()
        out = m(A, X)                      # [T, 3S]
        loss = crit(out, L[0])
        loss.backward()
        opt.step()
    tr = None
    for name in ("nn.Module + torch Adam", "TrainStep"):
        if name == "TrainStep":
            tr = TrainStep(m)
            fn = lambda: tr.step(A, X, L)
        else:
            fn = ref_style
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 100 * 1e6
        # what the step is made of: the library's kernels (hipEvent-bracketed, ~2 us of event latency each), the rest is
        # launch gaps + (module path) torch's own MSELoss / autograd / Adam kernels and host time
        _lib.profile_enable(True)
        for _ in range(20): fn()
        torch.cuda.synchronize()
        recs = sorted(_lib.profile_read(), key=lambda r: -r["ms"])
        _lib.profile_enable(False)
        ksum = sum(r["ms"] for r in recs) / 20 * 1e3
        print("%-6s %-24s %.0f us/step; library kernels %.0f us in %d launches/step: %s" % (
            math, name, wall, ksum, sum(r["launches"] for r in recs) // 20,
            "  ".join("%s=%.1f" % (r["name"].replace("_kernel", ""), r["ms"] / 20 * 1e3) for r in recs[:9])), flush=True)
