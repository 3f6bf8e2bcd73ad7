"""GPU parity: the HIP path (through the C ABI) against the golden fixtures and the CPU oracle.
Tolerances: Y <= 1e-4 max-abs; gradients <= 1e-4 relative to the tensor's max."""
import numpy as np
import pytest
import torch

from conftest import PARAM_KEYS, load_fixture, max_abs, rel_to_max

pytestmark = pytest.mark.gpu

Y_TOL = 1e-4
G_TOL = 1e-4


def _dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def _model_from(params, S, H, math="f32"):
    from windgnn_amd import GCN_GRU
    m = GCN_GRU(13, 13, 13, S * 13, H, math=math)
    m.load_state_dict({k: v.clone() for k, v in params.items()})   # reference keys, src/main.py:99
    return m.to(_dev())


def _run_step(model, A, X, L):
    """src/main.py:66-79 with the library's MSE op."""
    from windgnn_amd.functional import mse_loss_grad
    model.zero_grad()
    out = model(A, X)
    Y = out if out.dim() == 3 else out.unsqueeze(0)
    loss, dY = mse_loss_grad(Y, L)
    Y.backward(dY)
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    return out.detach().cpu(), float(loss), grads


@pytest.mark.parametrize("math", ["f32", "f16x3"])
def test_golden_forward_backward(golden, math):
    dev = _dev()
    S = golden["A"].shape[0]
    H = golden["Y"].shape[-1]
    model = _model_from(golden["params"], S, H, math)
    assert list(model.state_dict().keys()) == PARAM_KEYS
    A = torch.from_numpy(golden["A"]).to(dev)
    X = torch.from_numpy(golden["X"]).to(dev)
    L = torch.from_numpy(golden["L"]).to(dev)
    out, loss, grads = _run_step(model, A, X, L)
    B = X.shape[0]
    assert tuple(out.shape) == ((X.shape[1], H) if B == 1 else (B, X.shape[1], H))   # squeeze(0), step6:26
    assert max_abs(out.reshape(golden["Y"].shape), golden["Y"]) <= Y_TOL
    assert abs(loss - float(golden["loss"])) <= 1e-5
    for k in PARAM_KEYS:
        assert rel_to_max(grads[k], golden["grads"][k]) <= G_TOL, k


def test_inference_no_grad_matches(golden):
    dev = _dev()
    S, H = golden["A"].shape[0], golden["Y"].shape[-1]
    model = _model_from(golden["params"], S, H)
    with torch.no_grad():                                            # src/main.py:100-102
        out = model(torch.from_numpy(golden["A"]).to(dev), torch.from_numpy(golden["X"]).to(dev))
    assert max_abs(out.cpu().reshape(golden["Y"].shape), golden["Y"]) <= Y_TOL


@pytest.mark.parametrize("math", ["f32", "f16x3"])
@pytest.mark.parametrize("S,T,B,H", [(34, 24, 256, 102), (7, 12, 32, 21), (34, 24, 37, 102), (5, 3, 17, 9),
                                     (16, 4, 16, 48), (48, 2, 3, 33), (1, 1, 1, 1)])
def test_against_oracle_random(S, T, B, H, math):
    """BASELINE configs[1] (S=34,T=24,B=256 fp32), configs[0] shape, and ragged / edge shapes."""
    from oracle import windgnn_oracle as orc
    dev = _dev()
    g = torch.Generator().manual_seed(1000 + S * 7 + B)
    A = torch.rand(S, S, generator=g) / S + 0.01
    X = torch.rand(B, T, S, 13, generator=g)
    L = torch.rand(B, T, H, generator=g)
    p = orc.init_params(S, 13, H, seed=S + H)
    Yo, loss_o, go = orc.train_step(A.double(), X.double(), L.double(), {k: v.double() for k, v in p.items()})
    model = _model_from(p, S, H, math)
    out, loss, grads = _run_step(model, A.to(dev), X.to(dev), L.to(dev))
    assert max_abs(out.reshape(Yo.shape), Yo) <= Y_TOL
    assert abs(loss - float(loss_o)) <= 1e-5 * max(1.0, float(loss_o))
    for k in PARAM_KEYS:
        assert rel_to_max(grads[k], go[k]) <= G_TOL, k


def test_graph_conv_layer_module_with_input_grad():
    """GraphConvLayer alone (src/step5_gcn_layer_model.py), including dX for a stacked use."""
    from windgnn_amd import GraphConvLayer
    dev = _dev()
    torch.manual_seed(5)
    S = 34
    A = (torch.rand(S, S) / S + 0.01)
    X = torch.rand(1, 6, S, 13, requires_grad=True)
    layer = GraphConvLayer(13, 13)
    ref_out = torch.relu(torch.matmul(torch.matmul(A.double(), X.double()), layer.weight.double()) + layer.bias.double())
    dout = torch.rand_like(ref_out)
    gX, gW, gb = torch.autograd.grad(ref_out, [X, layer.weight, layer.bias], dout)
    layer_d = GraphConvLayer(13, 13).to(dev)
    layer_d.load_state_dict(layer.state_dict())
    Xd = X.detach().to(dev).requires_grad_(True)
    out = layer_d(A.to(dev), Xd)
    out.backward(dout.float().to(dev))
    assert max_abs(out.detach().cpu(), ref_out) <= 1e-5
    assert rel_to_max(Xd.grad.cpu(), gX) <= 1e-5
    assert rel_to_max(layer_d.weight.grad.cpu(), gW) <= 1e-5
    assert rel_to_max(layer_d.bias.grad.cpu(), gb) <= 1e-5


def test_adam_three_steps_match_reference():
    """src/main.py:52,80 on the flat-buffer Adam kernel, against the reference's parameters after
    1 and 3 optimiser steps on a fixed batch."""
    from windgnn_amd.functional import adam_step_
    dev = _dev()
    fx = load_fixture("f2b_s7_t12_b4_rand")
    S, H = 7, 21
    model = _model_from(fx["params"], S, H)
    A, X, L = (torch.from_numpy(fx[k]).to(dev) for k in ("A", "X", "L"))
    ps = list(model.parameters())
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for step in (1, 2, 3):
        _run_step(model, A, X, L)
        with torch.no_grad():
            for p, m, v in zip(ps, ms, vs):
                adam_step_(p.data, p.grad, m, v, step)
        if step in (1, 3):
            for k, p in model.named_parameters():
                assert max_abs(p.detach().cpu(), fx["a%d.%s" % (step, k)]) <= 2e-5, (step, k)


@pytest.mark.parametrize("math", ["f32", "f16x3"])
def test_bitwise_run_to_run_determinism(math):
    dev = _dev()
    fx = load_fixture("f3b_s34_t24_b4_rand")
    model = _model_from(fx["params"], 34, 102, math)
    A, X, L = (torch.from_numpy(fx[k]).to(dev) for k in ("A", "X", "L"))
    o1, l1, g1 = _run_step(model, A, X, L)
    o2, l2, g2 = _run_step(model, A, X, L)
    assert torch.equal(o1, o2) and l1 == l2
    for k in PARAM_KEYS:
        assert torch.equal(g1[k], g2[k]), k


def test_errors_are_loud():
    from windgnn_amd import GCN_GRU
    dev = _dev()
    big = GCN_GRU(13, 13, 13, 65 * 13, 33, math="f16x3").to(dev)
    with pytest.raises(RuntimeError, match="not supported"):     # > 64 stations: dense path not built, no fallback
        big(torch.rand(65, 65, device=dev), torch.rand(1, 2, 65, 13, device=dev))
    m = GCN_GRU(13, 13, 13, 34 * 13, 102).to(dev)
    with pytest.raises(RuntimeError):                      # CPU tensors: no fallback
        m(torch.rand(34, 34), torch.rand(1, 4, 34, 13))
    with pytest.raises(RuntimeError):                      # wrong station count (reference: .view fails)
        m(torch.rand(7, 7, device=dev), torch.rand(1, 4, 7, 13, device=dev))


def test_f16x3_tiny_gradients_survive_range_scaling():
    """dY ~ 1e-9 (a 4096-window batch mean) is far below fp16's range; the power-of-two scaling
    taken from max|dY| must keep the gradients at fp32-grade relative error."""
    from oracle import windgnn_oracle as orc
    from windgnn_amd.functional import gcn_gru
    dev = _dev()
    fx = load_fixture("f3b_s34_t24_b4_rand")
    A, X = torch.from_numpy(fx["A"]), torch.from_numpy(fx["X"])
    p = fx["params"]
    g = torch.Generator().manual_seed(3)
    dY = (torch.rand(4, 24, 102, generator=g) - 0.5) * 1e-9
    Yo, cache = orc.forward(A.double(), X.double(), {k: v.double() for k, v in p.items()})
    go = orc.backward(A.double(), X.double(), {k: v.double() for k, v in p.items()}, Yo, cache, dY.double())
    model = _model_from(p, 34, 102, "f16x3")
    Y = gcn_gru(A.to(dev), X.to(dev), model.hot_path_parameters(), model.math)
    Y.backward(dY.to(dev))
    for k, v in model.named_parameters():
        assert rel_to_max(v.grad.cpu(), go[k]) <= G_TOL, k


def test_make_windows_bit_exact_vs_oracle():
    """N2: on-device window batcher == src/step4_sequence_preparer.py:7-21 (oracle.make_windows), bit for bit."""
    import numpy as np
    from oracle import windgnn_oracle as orc
    from windgnn_amd.data import make_windows
    dev = _dev()
    rng = np.random.default_rng(5)
    Ttot, S, seq = 131, 7, 12
    data = rng.random((Ttot, S, 15)).astype(np.float32)          # 2 id columns + 13 features, as the reference
    xs, ys = orc.make_windows(data[: (Ttot - 3) // seq * seq + 3], seq)
    feat = torch.from_numpy(np.ascontiguousarray(data[:, :, 2:15])).to(dev)
    X, L = make_windows(feat, seq)
    assert X.shape == xs.shape and L.shape == ys.shape
    assert torch.equal(X.cpu(), torch.from_numpy(xs)) and torch.equal(L.cpu(), torch.from_numpy(ys))
    perm = [3, 0, 7, 5]                                            # shuffled windows (:23-26)
    Xp, Lp = make_windows(feat, seq, starts=[p * seq for p in perm])
    assert torch.equal(Xp.cpu(), torch.from_numpy(xs[perm])) and torch.equal(Lp.cpu(), torch.from_numpy(ys[perm]))
    with pytest.raises(RuntimeError):                              # a window whose labels run past the data
        make_windows(feat, seq, starts=[Ttot - seq - 2])


def test_predict_last_matches_reference_readout():
    """N4: last-timestep, de-normalised read-out of src/main.py:103,116."""
    from windgnn_amd.data import predict_last
    dev = _dev()
    fx = load_fixture("f2_s7_t12_b32_ckpt")
    model = _model_from(fx["params"], 7, 21)
    with torch.no_grad():
        Y = model(torch.from_numpy(fx["A"]).to(dev), torch.from_numpy(fx["X"]).to(dev))
    wmin, wmax = 0.0, 87.5
    out = predict_last(Y, wmin, wmax).cpu()
    ref = torch.from_numpy(fx["Y"])[:, -1, :] * (wmax - wmin) + wmin
    assert max_abs(out, ref) <= 1e-4 * (wmax - wmin)


def test_full_size_properties_B4096():
    """BASELINE's full size (S=34, T=24, B=4096, H=102, f16x3): properties that need no oracle run.
    (a) windows are independent: the big batch equals its two halves run separately, bit for bit;
    (b) the backward is linear in dY and the range scaling is a power of two: grads(4*dY) == 4*grads(dY) exactly;
    (c) gradients add over windows: grads(batch) ~= grads(half 1) + grads(half 2);
    (d) a 256-window slice agrees with the fp64 oracle."""
    from oracle import windgnn_oracle as orc
    from windgnn_amd.functional import gcn_gru_backward_raw, gcn_gru_forward_raw
    import numpy as np, os
    from conftest import GOLDEN
    dev = _dev()
    S, T, B, H = 34, 24, 4096, 102
    A = torch.from_numpy(np.load(os.path.join(GOLDEN, "graph_7_34.npz"))["A34"]).float()
    g = torch.Generator().manual_seed(99)
    X = torch.rand(B, T, S, 13, generator=g)
    dY = (torch.rand(B, T, H, generator=g) - 0.5) * 1e-6
    p = orc.init_params(S, 13, H, seed=3)
    model = _model_from(p, S, H, "f16x3")
    params = [q.detach() for q in model.hot_path_parameters()]
    Ad, Xd, dYd = A.to(dev), X.to(dev), dY.to(dev)

    def run(Xs, dYs):
        Y, stash, d = gcn_gru_forward_raw(Ad, Xs, params, model.math, want_stash=True)
        grads = [torch.empty_like(q) for q in params]
        gcn_gru_backward_raw(d, Ad, Xs, params, Y, dYs, stash, grads)
        return Y, grads

    Y, G = run(Xd, dYd)
    Y1, G1 = run(Xd[: B // 2].contiguous(), dYd[: B // 2].contiguous())
    Y2, G2 = run(Xd[B // 2:].contiguous(), dYd[B // 2:].contiguous())
    assert torch.equal(Y[: B // 2], Y1) and torch.equal(Y[B // 2:], Y2)                       # (a)
    _, G4 = run(Xd, (dYd * 4.0).contiguous())
    for a, b in zip(G, G4):
        assert torch.equal(a * 4.0, b)                                                        # (b)
    for a, b, c in zip(G, G1, G2):
        assert rel_to_max((b + c).cpu(), a.cpu()) <= 1e-5                                      # (c)
    n = 256
    Yo, cache = orc.forward(A.double(), X[:n].double(), {k: v.double() for k, v in p.items()})
    assert max_abs(Y[:n].cpu(), Yo) <= Y_TOL                                                   # (d)


def test_backward_in_two_parts_equals_one_call():
    """wgnn_bwd_part(1) then (2) must give bit-identical gradients to wgnn_bwd (the data-parallel overlap path)."""
    from windgnn_amd.functional import gcn_gru_backward_raw, gcn_gru_forward_raw
    dev = _dev()
    fx = load_fixture("f3b_s34_t24_b4_rand")
    model = _model_from(fx["params"], 34, 102, "f16x3")
    params = [q.detach() for q in model.hot_path_parameters()]
    A, X = torch.from_numpy(fx["A"]).to(dev), torch.from_numpy(fx["X"]).to(dev)
    dY = torch.from_numpy(fx["L"]).to(dev) * 1e-3
    Y, stash, d = gcn_gru_forward_raw(A, X, params, model.math, want_stash=True)
    g1 = [torch.zeros_like(q) for q in params]
    g2 = [torch.zeros_like(q) for q in params]
    gcn_gru_backward_raw(d, A, X, params, Y, dY, stash, g1, part=7)
    gcn_gru_backward_raw(d, A, X, params, Y, dY, stash, g2, part=1)
    gcn_gru_backward_raw(d, A, X, params, Y, dY, stash, g2, part=4)
    assert all(torch.equal(a, b) for a, b in zip(g1[4:], g2[4:]))        # GRU gradients final after parts 1+4
    gcn_gru_backward_raw(d, A, X, params, Y, dY, stash, g2, part=2)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))


# ---- "f16" math mode: plain fp16 operands, one MFMA pass (BASELINE configs[2], the 16-bit configuration).
# Its own tolerance, never reported as meeting the fp32 bar: fp16 has an 11-bit significand; observed max
# |Y - reference| is 1e-3 with the trained checkpoints and 8e-3 with N(0,1) random-init conv weights (plain
# bf16 measures 3e-2 / 7e-2 on the same fixtures); gradients within 5e-2 of their tensor's max.
F16_Y_TOL = 2e-2
F16_G_TOL = 5e-2


@pytest.mark.parametrize("fixture", ["f2_s7_t12_b32_ckpt", "f3_s34_t24_b4_ckpt", "f3b_s34_t24_b4_rand"])
def test_f16_mode_within_its_stated_tolerance(fixture):
    dev = _dev()
    fx = load_fixture(fixture)
    S, H = fx["A"].shape[0], fx["Y"].shape[-1]
    model = _model_from(fx["params"], S, H, "f16")
    A, X, L = (torch.from_numpy(fx[k]).to(dev) for k in ("A", "X", "L"))
    out, loss, grads = _run_step(model, A, X, L)
    ey = max_abs(out.reshape(fx["Y"].shape), fx["Y"])
    assert ey <= F16_Y_TOL, ey
    assert ey > 1e-6          # sanity: this really is the 16-bit path, not the split one
    for k in PARAM_KEYS:
        assert rel_to_max(grads[k], fx["grads"][k]) <= F16_G_TOL, k


# ---- general shapes: CSR adjacency (BASELINE configs[4]) and hidden widths beyond the fast kernels ----

def _oracle_step(A, X, L, p):
    from oracle import windgnn_oracle as orc
    return orc.train_step(A.double(), X.double(), L.double(), {k: v.double() for k, v in p.items()})


@pytest.mark.parametrize("math", ["f32", "f16x3"])
@pytest.mark.parametrize("S,T,B,H,k", [(34, 6, 5, 102, 4), (200, 3, 4, 60, 8), (300, 2, 3, 150, 8), (7, 4, 2, 21, 6)])
def test_csr_adjacency_against_oracle(S, T, B, H, k, math):
    """k-NN graph in CSR through wgnn_fwd / wgnn_bwd against the dense CPU oracle (S > 64 has no dense path)."""
    from oracle import windgnn_oracle as orc
    from windgnn_amd.graph import CsrAdjacency, build_knn_adjacency, synthetic_station_coords
    dev = _dev()
    csr = CsrAdjacency(*build_knn_adjacency(synthetic_station_coords(S, seed=S), k))
    A = csr.dense()
    g = torch.Generator().manual_seed(77 + S)
    X = torch.rand(B, T, S, 13, generator=g)
    L = torch.rand(B, T, H, generator=g)
    p = orc.init_params(S, 13, H, seed=S + H)
    Yo, loss_o, go = _oracle_step(A, X, L, p)
    model = _model_from(p, S, H, math)
    out, loss, grads = _run_step(model, csr.to(dev), X.to(dev), L.to(dev))
    assert max_abs(out.reshape(Yo.shape), Yo) <= Y_TOL
    assert abs(loss - float(loss_o)) <= 1e-5 * max(1.0, float(loss_o))
    for key in PARAM_KEYS:
        assert rel_to_max(grads[key], go[key]) <= G_TOL, key
    with torch.no_grad():                                    # inference: no stash (src/main.py:100-102)
        out2 = model(csr, X.to(dev))
    assert max_abs(out2.cpu().reshape(Yo.shape), Yo) <= Y_TOL


def test_csr_of_the_34_station_graph_matches_the_dense_path():
    """The same adjacency handed over dense and as CSR gives the same result (fp32 family, different kernels)."""
    from oracle import windgnn_oracle as orc
    from windgnn_amd.graph import CsrAdjacency
    dev = _dev()
    import os
    from conftest import GOLDEN
    A = torch.from_numpy(np.load(os.path.join(GOLDEN, "graph_7_34.npz"))["A34"]).float()   # reference build_graph output
    S, T, B, H = 34, 24, 8, 102
    g = torch.Generator().manual_seed(3)
    X = torch.rand(B, T, S, 13, generator=g)
    L = torch.rand(B, T, H, generator=g)
    p = orc.init_params(S, 13, H, seed=1)
    model = _model_from(p, S, H, "f32")
    out_d, loss_d, grads_d = _run_step(model, A.to(dev), X.to(dev), L.to(dev))
    out_c, loss_c, grads_c = _run_step(model, CsrAdjacency.from_dense(A).to(dev), X.to(dev), L.to(dev))
    assert max_abs(out_c, out_d) <= 1e-5
    for key in PARAM_KEYS:
        assert rel_to_max(grads_c[key], grads_d[key]) <= 1e-5, key


@pytest.mark.parametrize("math", ["f32", "f16x3"])
@pytest.mark.parametrize("S,T,B,H", [(20, 5, 6, 200), (34, 3, 4, 130)])
def test_wide_hidden_state_against_oracle(S, T, B, H, math):
    """gru_hidden_dim is a free constructor argument (src/step6_gcn_gru_combined_model.py:7): widths beyond the
    register-resident recurrence kernels take the per-step GEMM path (fp32 MFMA GEMM, or the plane GEMM in f16x3)."""
    from oracle import windgnn_oracle as orc
    dev = _dev()
    g = torch.Generator().manual_seed(500 + H)
    A = torch.rand(S, S, generator=g) / S + 0.01
    X = torch.rand(B, T, S, 13, generator=g)
    L = torch.rand(B, T, H, generator=g)
    p = orc.init_params(S, 13, H, seed=H)
    Yo, loss_o, go = _oracle_step(A, X, L, p)
    model = _model_from(p, S, H, math)
    out, loss, grads = _run_step(model, A.to(dev), X.to(dev), L.to(dev))
    assert max_abs(out.reshape(Yo.shape), Yo) <= Y_TOL
    for key in PARAM_KEYS:
        assert rel_to_max(grads[key], go[key]) <= G_TOL, key


def test_dense_adjacency_beyond_64_stations_is_refused():
    from oracle import windgnn_oracle as orc
    dev = _dev()
    S, H = 65, 12
    model = _model_from(orc.init_params(S, 13, H, seed=0), S, H, "f32")
    with pytest.raises(RuntimeError, match="not supported"):
        model(torch.rand(S, S).to(dev), torch.rand(1, 2, S, 13).to(dev))


@pytest.mark.parametrize("math", ["f32", "f16x3"])
def test_4096_station_csr_config_full_width_stations(math):
    """BASELINE configs[4] graph (4096-station symmetric 8-NN, CSR) at full S with a narrow GRU so that the CPU
    oracle stays small: exercises the SpMM kernels, the 53 248-wide input projection and its gradients."""
    from oracle import windgnn_oracle as orc
    from windgnn_amd.graph import CsrAdjacency, build_knn_adjacency, synthetic_station_coords
    dev = _dev()
    S, T, B, H = 4096, 2, 2, 24
    csr = CsrAdjacency(*build_knn_adjacency(synthetic_station_coords(S, seed=7), 8))
    A = csr.dense()
    g = torch.Generator().manual_seed(4096)
    X = torch.rand(B, T, S, 13, generator=g)
    L = torch.rand(B, T, H, generator=g)
    p = orc.init_params(S, 13, H, seed=9)
    # nn.GRU initialises with U(+-1/sqrt(hidden)); at the configuration's real width (12 288) the 53 248-term input
    # projection has the magnitude it has in that configuration.  Keep that scale here: with U(+-1/sqrt(24)) the
    # pre-activations reach the hundreds, every gate saturates and the gradients become ill-conditioned.
    p["gru.weight_ih_l0"] = p["gru.weight_ih_l0"] * (H / 12288.0) ** 0.5
    Yo, loss_o, go = _oracle_step(A, X, L, p)
    model = _model_from(p, S, H, math)
    out, loss, grads = _run_step(model, csr.to(dev), X.to(dev), L.to(dev))
    assert max_abs(out.reshape(Yo.shape), Yo) <= Y_TOL
    for key in PARAM_KEYS:
        assert rel_to_max(grads[key], go[key]) <= G_TOL, key


@pytest.mark.parametrize("S,T,B,H,math", [(34, 24, 37, 102, "f16x3"), (34, 24, 37, 102, "f16"), (7, 12, 5, 21, "f16x3"),
                                          (34, 6, 9, 102, "f32"), (20, 4, 6, 200, "f16x3"), (1, 1, 1, 1, "f16x3")])
def test_fused_loss_backward_equals_loss_then_backward(S, T, B, H, math):
    """wgnn_bwd_mse_part (what TrainStep calls) against wgnn_mse_loss_grad + wgnn_bwd and against the oracle, for
    the kernel that forms dY from the labels itself (f16x3 / f16, H <= 127) and for the shapes that build dY."""
    from oracle import windgnn_oracle as orc
    from windgnn_amd.functional import gcn_gru_backward_mse_raw, gcn_gru_forward_raw
    dev = _dev()
    g = torch.Generator().manual_seed(31 + S + H)
    A = torch.rand(S, S, generator=g) / S + 0.01
    X = torch.rand(B, T, S, 13, generator=g)
    L = torch.rand(B, T, H, generator=g)
    p = orc.init_params(S, 13, H, seed=S)
    Yo, loss_o, go = _oracle_step(A, X, L, p)
    model = _model_from(p, S, H, math)
    out, loss_u, grads_u = _run_step(model, A.to(dev), X.to(dev), L.to(dev))          # loss op, then backward
    params = [q.detach() for q in model.hot_path_parameters()]
    Y, stash, d = gcn_gru_forward_raw(A.to(dev), X.to(dev), params, model.math, want_stash=True)
    grads = [torch.empty_like(q) for q in params]
    loss = torch.empty((), device=dev)
    gcn_gru_backward_mse_raw(d, A.to(dev), X.to(dev), params, Y, L.to(dev), stash, grads, loss, 1.0)
    tol = 5e-2 if math == "f16" else G_TOL
    assert abs(float(loss) - float(loss_o)) <= (2e-3 if math == "f16" else 1e-5) * max(1.0, float(loss_o))
    for key, gf in zip(PARAM_KEYS, grads):
        assert rel_to_max(gf.cpu(), go[key]) <= tol, key
        assert rel_to_max(gf.cpu(), grads_u[key]) <= (1e-3 if math == "f16" else 2e-5), key
    # in two parts, with a gradient scale (the data-parallel call pattern)
    grads2 = [torch.empty_like(q) for q in params]
    gcn_gru_backward_mse_raw(d, A.to(dev), X.to(dev), params, Y, L.to(dev), stash, grads2, loss, 0.5, part=1 | 4)
    gcn_gru_backward_mse_raw(d, A.to(dev), X.to(dev), params, Y, L.to(dev), stash, grads2, loss, 0.5, part=2)
    for a, b in zip(grads, grads2):
        assert rel_to_max(2.0 * b.cpu(), a.cpu()) <= 1e-6


def test_graph_conv_layer_with_csr_adjacency():
    """GraphConvLayer alone (src/step5_gcn_layer_model.py) over a 150-station k-NN graph in CSR, including dX."""
    from windgnn_amd import GraphConvLayer
    from windgnn_amd.graph import CsrAdjacency, build_knn_adjacency, synthetic_station_coords
    dev = _dev()
    torch.manual_seed(6)
    S = 150
    csr = CsrAdjacency(*build_knn_adjacency(synthetic_station_coords(S, seed=2), 6))
    A = csr.dense()
    X = torch.rand(2, 3, S, 13, requires_grad=True)
    layer = GraphConvLayer(13, 13)
    ref_out = torch.relu(torch.matmul(torch.matmul(A.double(), X.double()), layer.weight.double()) + layer.bias.double())
    dout = torch.rand_like(ref_out)
    gX, gW, gb = torch.autograd.grad(ref_out, [X, layer.weight, layer.bias], dout)
    layer_d = GraphConvLayer(13, 13).to(dev)
    layer_d.load_state_dict(layer.state_dict())
    Xd = X.detach().to(dev).requires_grad_(True)
    out = layer_d(csr.to(dev), Xd)
    out.backward(dout.float().to(dev))
    assert max_abs(out.detach().cpu(), ref_out) <= 1e-5
    assert rel_to_max(Xd.grad.cpu(), gX) <= 1e-5
    assert rel_to_max(layer_d.weight.grad.cpu(), gW) <= 1e-5
    assert rel_to_max(layer_d.bias.grad.cpu(), gb) <= 1e-5
