"""Does running the two independent backward chains (weight-gradient GEMMs | dg + GCN backward) on two streams pay?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU
from windgnn_amd.functional import gcn_gru_backward_raw, gcn_gru_forward_raw, mse_loss_grad
dev = torch.device("cuda:0")
m = GCN_GRU(F, F, F, S * F, H, math="f16x3").to(dev)
params = [p.detach() for p in m.hot_path_parameters()]
grads = [torch.zeros_like(p) for p in params]
A = adjacency_34().to(dev)
X, L = make_inputs(4096, 0, dev)
Y, stash, d = gcn_gru_forward_raw(A, X, params, m.math, want_stash=True)
loss, dY = mse_loss_grad(Y, L)
side = torch.cuda.Stream()
def seq():
    gcn_gru_backward_raw(d, A, X, params, Y, dY, stash, grads, part=7)
def par():
    gcn_gru_backward_raw(d, A, X, params, Y, dY, stash, grads, part=1)
    ev = torch.cuda.Event(); ev.record()
    side.wait_event(ev)
    gcn_gru_backward_raw(d, A, X, params, Y, dY, stash, grads, part=4, stream=side)
    gcn_gru_backward_raw(d, A, X, params, Y, dY, stash, grads, part=2)
    ev2 = torch.cuda.Event(); ev2.record(side)
    torch.cuda.current_stream().wait_event(ev2)
for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print("%-12s backward %.1f us" % (name, (time.perf_counter() - t0) / 50 * 1e6))
