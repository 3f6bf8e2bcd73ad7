"""Pin the CPU oracle to the reference: every golden fixture (produced by the reference itself,
oracle/make_golden.py) must be reproduced by the restatement in oracle/windgnn_oracle.py."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PARAM_KEYS, WINDOW_FIXTURES, load_fixture, max_abs, rel_to_max
from oracle import windgnn_oracle as orc


def test_forward_loss_matches_reference(golden):
    A, X, L = (torch.from_numpy(golden[k]) for k in ("A", "X", "L"))
    Y, _ = orc.forward(A, X, golden["params"])
    assert Y.shape == golden["Y"].shape
    assert max_abs(Y, golden["Y"]) <= 1e-5                      # restatement vs import
    loss, _ = orc.mse_loss_and_grad(Y, L)
    assert abs(float(loss) - float(golden["loss"])) <= 1e-6


def test_backward_matches_reference_autograd(golden):
    A, X, L = (torch.from_numpy(golden[k]) for k in ("A", "X", "L"))
    _, _, grads = orc.train_step(A, X, L, golden["params"])
    for k in PARAM_KEYS:
        assert grads[k].shape == golden["grads"][k].shape, k
        assert rel_to_max(grads[k], golden["grads"][k]) <= 2e-5, k


def test_backward_fp64_tight(golden):
    """In float64 the hand-derived backward and the reference's fp32 autograd differ only by
    the reference's own rounding."""
    A = torch.from_numpy(golden["A"]).double()
    X = torch.from_numpy(golden["X"]).double()
    L = torch.from_numpy(golden["L"]).double()
    p = {k: v.double() for k, v in golden["params"].items()}
    _, _, grads = orc.train_step(A, X, L, p)
    for k in PARAM_KEYS:
        assert rel_to_max(grads[k], golden["grads"][k]) <= 2e-5, k


def test_adam_steps_match_reference(golden):
    A, X, L = (torch.from_numpy(golden[k]) for k in ("A", "X", "L"))
    p = {k: v.clone() for k, v in golden["params"].items()}
    st = orc.adam_init(p)
    for step in (1, 2, 3):
        _, _, grads = orc.train_step(A, X, L, p)
        p = orc.adam_step(p, grads, st)
        if step in (1, 3):
            for k in PARAM_KEYS:
                # one Adam step moves every weight by ~lr=1e-3; agreement to 2e-5 pins sign+size
                assert max_abs(p[k], golden["a%d.%s" % (step, k)]) <= 2e-5, (step, k)


def test_b1_squeeze_shape():
    fx = load_fixture("f1_tiny_s3_t2_b1")
    Y, _ = orc.forward(torch.from_numpy(fx["A"]), torch.from_numpy(fx["X"]), fx["params"], want_cache=False)
    assert orc.reference_output_shape(Y).shape == (2, 9)


def test_build_graph_matches_reference():
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "graph_7_34.npz"))
    A34 = orc.build_graph(z["coords34"])
    assert np.abs(A34 - z["A34"]).max() <= 1e-12
    A7 = orc.build_graph(z["coords34"][:7])
    assert np.abs(A7 - z["A7"]).max() <= 1e-12


@pytest.mark.parametrize("name", WINDOW_FIXTURES)
def test_make_windows_matches_reference_create_sequences(name):
    """Row N2 pin: oracle.make_windows (no shuffle) + the stored permutation == what the reference's
    __create_sequences returned for the same array (src/step4_sequence_preparer.py:7-27), bit for bit."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    xs, ys = orc.make_windows(z["data"], int(z["seq"]))
    assert xs.shape == z["xs"].shape and ys.shape == z["ys"].shape
    assert np.array_equal(xs[z["perm"]], z["xs"]) and np.array_equal(ys[z["perm"]], z["ys"])
    assert sorted(z["perm"].tolist()) == list(range(xs.shape[0]))
