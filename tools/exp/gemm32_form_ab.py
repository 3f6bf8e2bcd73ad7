"""Exact-fp32 step at B = 4096: the two NT products (GI, dg) as one 8-wave workgroup per CU (form 0) or two
4-wave workgroups per CU with the second of a pair started late (WGNN_OPT_GEMM32_FORM = 1 + stagger units), same process.
Checks that Y and the eight gradients of a step are bit-identical between the forms, then times the step and its kernels.
    python tools/exp/gemm32_form_ab.py [forms ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
forms = [int(a) for a in sys.argv[1:]] or [0, 1, 3, 5, 7, 9, 11, 13, 17, 0]
A = adjacency_34().to(dev)
X, L = make_inputs(4096, 0, dev)


def grads_of(form):
    _lib.set_option(_lib.OPT_GEMM32_FORM, form)
    torch.manual_seed(5)
    m = GCN_GRU(F, F, F, S * F, H, math="f32").to(dev)
    out = m(A, X)
    loss = torch.nn.functional.mse_loss(out, L)
    loss.backward()
    return [out.detach().clone()] + [p.grad.clone() for p in m.parameters()]


ref = grads_of(0)
for form in (1, 4, 34):
    got = grads_of(form)
    print("form %d: Y and 8 gradients bitwise equal to form 0: %s" % (form, all(torch.equal(a, b) for a, b in zip(ref, got))), flush=True)

m = GCN_GRU(F, F, F, S * F, H, math="f32").to(dev)
tr = TrainStep(m)
step = lambda: tr.step(A, X, L)
for _ in range(10):
    step()
for rnd in range(2):
    for form in forms:
        _lib.set_option(_lib.OPT_GEMM32_FORM, form)
        for _ in range(5):
            step()
        _lib.profile_enable(True)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        recs = _lib.profile_read()
        _lib.profile_enable(False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            step()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40
        ks = "  ".join("%s=%.1f" % (r["name"].replace("_kernel", ""), 1e3 * r["ms"] / r["launches"]) for r in recs if "gemm32_nt" in r["name"])
        print("round %d form %2d  step %.1f us = %.3f M windows/s  [%s]" % (rnd, form, us, 4096 / us, ks), flush=True)
_lib.set_option(_lib.OPT_GEMM32_FORM, 0)
