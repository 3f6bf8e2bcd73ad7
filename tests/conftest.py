import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
PARAM_KEYS = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias",
              "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0"]
FIXTURES = ["f1_tiny_s3_t2_b1", "f1b_tiny_s3_t5_b3_h5", "f2_s7_t12_b32_ckpt", "f2b_s7_t12_b4_rand",
            "f3_s34_t24_b4_ckpt", "f3b_s34_t24_b4_rand", "f4_s34_t168_b1_ckpt", "f5_s7_t12_b1_rand"]
WINDOW_FIXTURES = ["w1_t131_s7_seq12", "w2_t75_s3_seq24"]   # outputs of the reference's __create_sequences


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    fx = {k: z[k] for k in z.files}
    fx["params"] = {k: torch.from_numpy(fx["p." + k]) for k in PARAM_KEYS}
    fx["grads"] = {k: torch.from_numpy(fx["g." + k]) for k in PARAM_KEYS}
    return fx


@pytest.fixture(params=FIXTURES)
def golden(request):
    return load_fixture(request.param)


def max_abs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def rel_to_max(a, b):
    b = torch.as_tensor(b).double()
    return max_abs(a, b) / max(float(b.abs().max()), 1e-30)
