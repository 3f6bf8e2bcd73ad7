"""Experiment: does overlapping the forward's three stages across batch chunks pay?  The forward of B = 4096
windows as ONE call on one stream, against the same windows as 2 (4) independent half (quarter) batches on 2 (4)
streams, so that one chunk's projection GEMM (MFMA / L2-bound) can run beside another chunk's GCN (VALU-bound) and
recurrence (bandwidth-bound).  Prints the time per 4096 windows for each arrangement."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import adjacency_34, make_inputs
from windgnn_amd import GCN_GRU
from windgnn_amd.functional import gcn_gru_forward_raw

dev = torch.device("cuda:0")
B = 4096
m = GCN_GRU(13, 13, 13, 34 * 13, 102, math="f16x3").to(dev)
P = [p.detach() for p in m.hot_path_parameters()]
A = adjacency_34().to(dev)
X, L = make_inputs(B, 0, dev)
for nchunk in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(nchunk)]
    Xs = [c.contiguous() for c in X.chunk(nchunk)]
    def run():
        for st, xc in zip(streams, Xs):
            with torch.cuda.stream(st):
                gcn_gru_forward_raw(A, xc, P, m.math, want_stash=True)
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    print("chunks/streams %d: %.1f us per 4096 windows" % (nchunk, 1e6 * (time.perf_counter() - t0) / n))
