"""Gradient / output errors of the HIP path against the fp64 oracle at S=34, T=24, H=102 (GPU box).
    [MATH=f16x3] [BS=256,1100] python tests/grad_error_probe.py
Prints, per batch size, max|Y - oracle| and per parameter the error relative to the tensor's max."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import windgnn_oracle as orc
from windgnn_amd import GCN_GRU
from windgnn_amd.functional import mse_loss_grad
from bench import adjacency_34
dev = torch.device("cuda:0")
math = os.environ.get("MATH", "f16x3")
KEYS = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0"]
torch.set_num_threads(16)
for B in [int(b) for b in os.environ.get("BS", "256,1100").split(",")]:
    S, T, H = 34, 24, 102
    g = torch.Generator().manual_seed(B)
    A = adjacency_34()
    X = torch.rand(B, T, S, 13, generator=g)
    L = torch.rand(B, T, H, generator=g)
    p = orc.init_params(S, 13, H, seed=3)
    Yo, loss_o, go = orc.train_step(A.double(), X.double(), L.double(), {k: v.double() for k, v in p.items()})
    m = GCN_GRU(13, 13, 13, S * 13, H, math=math).to(dev)
    m.load_state_dict(p)
    Y = m(A.to(dev), X.to(dev))
    loss, dY = mse_loss_grad(Y, L.to(dev))
    Y.backward(dY)
    out = "B=%d %s  Y %.2e " % (B, math, float((Y.detach().cpu().double() - Yo).abs().max()))
    for k, v in m.named_parameters():
        e = float((v.grad.cpu().double() - go[k]).abs().max()) / float(go[k].abs().max())
        out += " %s %.2e" % (k.replace("gru.", "").replace("weight", "w").replace("bias", "b"), e)
    print(out, flush=True)
