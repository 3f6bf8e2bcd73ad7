"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference/src).

Runs only in the build container (the reference does not travel to the GPU box); the fixtures
it writes are data (inputs + expected outputs) and are committed.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Each fixture holds: A, X, L, the 8 parameters (p.<key>), Y, loss, the 8 gradients (g.<key>) and
the parameters after 1 and 3 Adam steps on the same batch (a1.<key>, a3.<key>), all produced by
reference code: GCN_GRU.forward (src/step6_gcn_gru_combined_model.py:13-27), nn.MSELoss and
torch.optim.Adam exactly as src/main.py:49-52,66-80 uses them.  For B > 1 the golden is the
stack of B independent reference calls (the reference's .view(1, ...) only accepts B == 1) and
the loss is the mean of the per-window losses.
"""
import os
import sys
import warnings

import numpy as np
import pandas as pd
import torch

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "src"))
from step6_gcn_gru_combined_model import GCN_GRU  # noqa: E402
from step2_graph_builder import build_graph  # noqa: E402
import step4_sequence_preparer as step4  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
F = 13
KEYS = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias",
        "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0"]


def station_frame(n):
    df = pd.read_csv(os.path.join(REF, "data", "ACISStationCoordinates.csv"))
    df = df[df["Station Name"] != "Enchant 2 AGCM"]          # src/step1_loading_preprocessing.py:32
    return df.iloc[:n].reset_index(drop=True)


def ref_adjacency(n):
    return np.asarray(build_graph(station_frame(n)), dtype=np.float64)


def run_reference(model, A, X, L):
    """Forward+loss+backward for a batch as B independent reference calls."""
    lossf = torch.nn.MSELoss()
    model.zero_grad()
    outs, total = [], 0.0
    B = X.shape[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for b in range(B):
            out = model(A, X[b:b + 1])                       # src/main.py:66
            outs.append(out)
            total = total + lossf(out, L[b:b + 1])           # src/main.py:72 (broadcast [T,H] vs [1,T,H])
    loss = total / B
    loss.backward()                                          # src/main.py:79
    Y = torch.stack([o.detach() for o in outs])
    grads = {k: v.grad.detach().clone() for k, v in model.named_parameters()}
    return Y, loss.detach(), grads


def make(name, S, T, B, seed, A64, state_dict=None, H=None, dims=None):
    """dims = (input_dim, hidden_dim): the rest of the constructor's domain (src/step6_gcn_gru_combined_model.py:7-11; output_dim
    stays 13, which :16 hard-codes); None = the reference's own 13 / 13 (src/main.py:41)."""
    torch.manual_seed(seed)
    H = 3 * S if H is None else H
    Fi, Fh = dims if dims is not None else (F, F)
    model = GCN_GRU(Fi, Fh, F, S * F, H)                     # src/main.py:41-42
    if state_dict is not None:
        model.load_state_dict(state_dict)                    # src/main.py:99
    A = torch.tensor(A64).float()                            # src/main.py:26
    g = torch.Generator().manual_seed(seed + 1)
    X = torch.rand(B, T, S, Fi, generator=g)
    L = torch.rand(B, T, H, generator=g)
    if dims is not None:                                     # randn conv weights of other widths: keep the activations O(1)
        with torch.no_grad():
            model.conv1.weight.mul_(0.5)
            model.conv2.weight.mul_(0.5)
    p0 = {k: v.detach().clone() for k, v in model.named_parameters()}
    Y, loss, grads = run_reference(model, A, X, L)
    fx = {"A": A.numpy(), "A64": A64, "X": X.numpy(), "L": L.numpy(), "Y": Y.numpy(),
          "loss": np.float32(loss.item())}
    for k in KEYS:
        fx["p." + k] = p0[k].numpy()
        fx["g." + k] = grads[k].numpy()
    opt = torch.optim.Adam(model.parameters(), lr=0.001)     # src/main.py:52
    for step in (1, 2, 3):
        run_reference(model, A, X, L)
        opt.step()                                           # src/main.py:80
        if step in (1, 3):
            for k, v in model.named_parameters():
                fx["a%d.%s" % (step, k)] = v.detach().clone().numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **fx)
    print("%-28s S=%d T=%d B=%d H=%d loss=%.6f  %.1f KB" % (name, S, T, B, H, loss.item(),
                                                            os.path.getsize(path) / 1024))


def make_layer(name, S, lead, Fi, Fo, seed, A64):
    """One reference GraphConvLayer(Fi, Fo) (src/step5_gcn_layer_model.py:5-23) with input gradient: X [*lead, S, Fi] requires
    grad, upstream gradient random: out, dW, db, dX."""
    from step5_gcn_layer_model import GraphConvLayer
    torch.manual_seed(seed)
    layer = GraphConvLayer(Fi, Fo)
    with torch.no_grad():
        layer.bias.uniform_(-0.5, 0.5)                       # the reference initialises it to zeros (:10): a trained value here
    A = torch.tensor(A64).float()
    g = torch.Generator().manual_seed(seed + 1)
    X = (torch.rand(*lead, S, Fi, generator=g) - 0.3).requires_grad_(True)
    out = layer(A, X)
    dout = torch.rand(out.shape, generator=g) - 0.5
    out.backward(dout)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, A=A.numpy(), X=X.detach().numpy(), W=layer.weight.detach().numpy(), b=layer.bias.detach().numpy(),
                        out=out.detach().numpy(), dout=dout.numpy(), dW=layer.weight.grad.numpy(), db=layer.bias.grad.numpy(),
                        dX=X.grad.numpy())
    print("%-28s S=%d lead=%s %d -> %d  %.1f KB" % (name, S, lead, Fi, Fo, os.path.getsize(path) / 1024))


def make_windows_fixture(name, Ttot, S, seq, seed):
    """Row N2: the reference's own window builder (src/step4_sequence_preparer.py:7-27, `__create_sequences`,
    a module-level function reached through the module dict) on a seeded [time, station, 15-column] array, with
    np.random seeded so its shuffle (:23-26) is reproducible.  Stored: the input array, the shuffled xs / ys it
    returned, and the permutation it drew (recovered by re-drawing with the same seed)."""
    create = step4.__dict__["__create_sequences"]
    rng = np.random.default_rng(seed)
    data = rng.random((Ttot, S, 15)).astype(np.float32)
    np.random.seed(seed)
    xs, ys = create(data, seq)
    np.random.seed(seed)
    perm = np.arange(xs.shape[0])
    np.random.shuffle(perm)                                   # the same draw as :23-24
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, data=data, xs=xs, ys=ys, perm=perm.astype(np.int32), seq=np.int32(seq))
    print("%-28s Ttot=%d S=%d seq=%d windows=%d  %.1f KB" % (name, Ttot, S, seq, xs.shape[0],
                                                              os.path.getsize(path) / 1024))


def main():
    only = set(sys.argv[1:])                                  # optional: regenerate just the named fixtures
    os.makedirs(OUT, exist_ok=True)
    if only:
        global make, make_windows_fixture, make_layer
        _make, _mw, _ml = make, make_windows_fixture, make_layer
        make = lambda name, *a, **k: _make(name, *a, **k) if name in only else None                  # noqa: E731
        make_windows_fixture = lambda name, *a, **k: _mw(name, *a, **k) if name in only else None   # noqa: E731
        make_layer = lambda name, *a, **k: _ml(name, *a, **k) if name in only else None             # noqa: E731
    A7, A34 = ref_adjacency(7), ref_adjacency(34)
    rng = np.random.default_rng(3)
    A3 = rng.random((3, 3)) * 0.5 + 0.05                      # tiny, deliberately asymmetric
    sd7 = torch.load(os.path.join(REF, "wind_gnn_7.pth"), map_location="cpu")
    sd34 = torch.load(os.path.join(REF, "wind_gnn_34.pth"), map_location="cpu")
    make("f1_tiny_s3_t2_b1", 3, 2, 1, 11, A3)
    make("f1b_tiny_s3_t5_b3_h5", 3, 5, 3, 12, A3, H=5)
    make("f2_s7_t12_b32_ckpt", 7, 12, 32, 21, A7, sd7)
    make("f2b_s7_t12_b4_rand", 7, 12, 4, 22, A7)
    make("f3_s34_t24_b4_ckpt", 34, 24, 4, 31, A34, sd34)
    make("f3b_s34_t24_b4_rand", 34, 24, 4, 32, A34)
    make("f4_s34_t168_b1_ckpt", 34, 168, 1, 41, A34, sd34)
    # B = 1 at the 7-station shape: the reference's own call shape (src/main.py:64-80), for the drop-in loop test
    make("f5_s7_t12_b1_rand", 7, 12, 1, 51, A7)
    # round 5: the constructor's domain beyond 13 / 13 (VERDICT r4 missing 3): GCN_GRU(input_dim, hidden_dim, 13, ...) and single
    # GraphConvLayer(in, out) layers with input gradients
    make("f6_s7_t12_b4_in5_hid20", 7, 12, 4, 61, A7, dims=(5, 20))
    make("f7_s34_t6_b2_in13_hid32", 34, 6, 2, 62, A34, dims=(13, 32))
    make("f8_s3_t2_b1_in64_hid1", 3, 2, 1, 63, A3, dims=(64, 1))
    make_layer("l1_s7_in6_out9", 7, (2, 5), 6, 9, 71, A7)
    make_layer("l2_s34_in13_out40", 34, (3,), 13, 40, 72, A34)
    make_layer("l3_s3_in64_out64", 3, (1, 4), 64, 64, 73, A3)
    # row N2: windows built by the reference's __create_sequences (src/step4_sequence_preparer.py:7-27)
    make_windows_fixture("w1_t131_s7_seq12", 131, 7, 12, 5)
    make_windows_fixture("w2_t75_s3_seq24", 75, 3, 24, 6)     # len % seq == 3: the last window's labels just fit
    if only and "graph_7_34" not in only:
        return
    # adjacency-only fixtures for the build_graph restatement (row N1)
    np.savez_compressed(os.path.join(OUT, "graph_7_34.npz"), A7=A7, A34=A34,
                        coords34=station_frame(34)[["Latitude", "Longitude"]].values)


if __name__ == "__main__":
    main()
