import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes as C
from windgnn_amd import _lib
from windgnn_amd.functional import finish_step, gcn_gru_backward_mse_raw, gcn_gru_forward_raw, prepared_weights
dev = torch.device("cuda:0")
S, T, B, H = 20, 4, 6, 200
g = torch.Generator().manual_seed(31 * S + H)
A = (torch.rand(S, S, generator=g) / S + 0.01).to(dev)
X = torch.rand(B, T, S, 13, generator=g).to(dev)
L = torch.rand(B, T, H, generator=g).to(dev)
k = 1.0 / H ** 0.5
shapes = [(13, 13), (13,), (13, 13), (13,), (3 * H, S * 13), (3 * H, H), (3 * H,), (3 * H,)]
ps = [(torch.rand(sh, generator=g) * 2 * k - k).to(dev) for sh in shapes]
gs = [torch.zeros_like(q) for q in ps]
ms = [torch.zeros_like(q) for q in ps]
vs = [torch.zeros_like(q) for q in ps]
mode = _lib.MATH_F16X3
loss = torch.zeros((), device=dev)
Y, stash, d = gcn_gru_forward_raw(A, X, ps, mode, labels=L)
img = prepared_weights(d, ps, dev)
Y, stash, d = gcn_gru_forward_raw(A, X, ps, mode, labels=L, prepared=img)
gcn_gru_backward_mse_raw(d, A, X, ps, Y, L, stash, gs, loss, 1.0, part=7 | 8 | _lib.BWD_DEFER, prepared=img)
finish_step(d, ps, gs, 6, dict(exp_avg=ms, exp_avg_sq=vs, step=1, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8), img)
again = prepared_weights(d, ps, dev)
a = img.view(torch.int16).cpu(); b = again.view(torch.int16).cpu()
bad = (a != b).nonzero().flatten()
print("mismatches:", bad.numel(), "of", a.numel())
I, G3 = S * 13, 3 * H
Ip = (I + 1 + 31) // 32 * 32
Gp = (G3 + 31) // 32 * 32
np_g3 = (G3 + 31) // 32 * 32 + 448 - 32
np_i = (I + 31) // 32 * 32 + 448 - 32
fsz = np_g3 * Ip            # halfs per plane, forward image
def al(x): return (x + 63) // 64 * 64
prep_b = al(np_g3 * Ip) * 2     # float offset -> halfs
print("fwd plane halfs", fsz, "bwd image starts at half", prep_b, "bwd plane halfs", np_i * Gp)
for idx in bad[:24].tolist():
    if idx < prep_b:
        plane, o = divmod(idx, fsz)
        kt, rem = divmod(o, np_g3 * 32); row, c = divmod(rem, 32)
        print("fwd plane", plane, "row", row, "col", kt * 32 + c, "got", a[idx].item(), "want", b[idx].item())
    else:
        o = idx - prep_b
        plane, o = divmod(o, np_i * Gp)
        kt, rem = divmod(o, np_i * 32); col, r = divmod(rem, 32)
        print("bwd plane", plane, "row(W)", kt * 32 + r, "col(W)", col, "got", a[idx].item(), "want", b[idx].item())
