"""Prints what the two trajectory tests observe (GPU box): per math mode, the worst |param - reference| after 1 / 3
optimiser steps of the drop-in loop (fixtures f5 / f1 / f4) and after 20 TrainStep steps (f2b), with the share of
elements beyond the tight bound.  The numbers pinned in tests/test_gpu_parity.py come from this script."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.nn as nn
from conftest import load_fixture
from windgnn_amd import GCN_GRU
from windgnn_amd.trainer import TrainStep
from oracle import windgnn_oracle as orc

dev = torch.device("cuda:0")
for math in ("f32", "f16x3"):
    for fixture in ("f5_s7_t12_b1_rand", "f1_tiny_s3_t2_b1", "f4_s34_t168_b1_ckpt"):
        fx = load_fixture(fixture)
        S, H = fx["A"].shape[0], fx["Y"].shape[-1]
        model = GCN_GRU(13, 13, 13, 13 * S, H, math=math)
        model.load_state_dict({k: v.clone() for k, v in fx["params"].items()})
        model = model.to(dev)
        A = torch.tensor(fx["A64"]).float().to(dev)
        lossf, opt = nn.MSELoss(), torch.optim.Adam(model.parameters(), lr=0.001)
        X, L = torch.from_numpy(fx["X"]).to(dev), torch.from_numpy(fx["L"]).to(dev)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for it in (1, 2, 3):
                out = model(A, X)
                opt.zero_grad()
                lossf(out, L).backward()
                opt.step()
                if it in (1, 3):
                    for k, v in model.state_dict().items():
                        err = (v.cpu().double() - torch.from_numpy(fx["a%d.%s" % (it, k)]).double()).abs()
                        if float(err.max()) > 1e-5:
                            print("dropin %-5s %-22s it=%d %-20s max %.3e  frac>2e-5 %.4f" % (math, fixture, it, k, float(err.max()), float((err > 2e-5).double().mean())))
    fx = load_fixture("f2b_s7_t12_b4_rand")
    A, X, L = (torch.from_numpy(fx[k]) for k in ("A", "X", "L"))
    p = {k: v.double() for k, v in fx["params"].items()}
    st = orc.adam_init(p)
    model = GCN_GRU(13, 13, 13, 91, 21, math=math)
    model.load_state_dict({k: v.clone() for k, v in fx["params"].items()})
    tr = TrainStep(model.to(dev), check_every=5)
    Ad, Xd, Ld = A.to(dev), X.to(dev), L.to(dev)
    for step in range(20):
        _, loss_o, g = orc.train_step(A.double(), X.double(), L.double(), p)
        p = orc.adam_step(p, g, st)
        loss, _ = tr.step(Ad, Xd, Ld)
    for k, v in model.named_parameters():
        err = (v.detach().cpu().double() - p[k]).abs()
        print("20step %-5s %-20s max %.3e  frac>1e-4 %.4f  frac>2e-5 %.4f" % (math, k, float(err.max()), float((err > 1e-4).double().mean()), float((err > 2e-5).double().mean())))
