"""Drop-in nn.Modules with the reference's names, constructor arguments, forward signatures and
state_dict keys, backed by the HIP library.

  GraphConvLayer(input_dim, output_dim)                         src/step5_gcn_layer_model.py:5-23
  GCN_GRU(input_dim, hidden_dim, output_dim, gru_input,
          gru_hidden_dim).forward(adj_matrix, attr_matrix)      src/step6_gcn_gru_combined_model.py:6-27
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib
from .functional import GraphConvFunction, check_range_status, gcn_gru

NUM_FEATURES = 13   # hard-coded in the reference: src/step6_gcn_gru_combined_model.py:16


class GraphConvLayer(nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        if input_dim != output_dim:
            raise RuntimeError("GraphConvLayer: the reference only ever uses input_dim == output_dim "
                               "(src/main.py:41); got %d -> %d" % (input_dim, output_dim))
        self.weight = nn.Parameter(torch.randn(input_dim, output_dim))   # step5:8-9
        self.bias = nn.Parameter(torch.zeros(output_dim))                # step5:10

    def forward(self, adj_matrix, attr_matrix):
        return GraphConvFunction.apply(adj_matrix, attr_matrix, self.weight, self.bias)


class _GRUParams(nn.Module):
    """Holds the four nn.GRU tensors under the reference's names (weight_ih_l0 ...), with
    nn.GRU's U(-1/sqrt(H), 1/sqrt(H)) initialisation."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / math.sqrt(hidden_size)
        self.weight_ih_l0 = nn.Parameter(torch.empty(3 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh_l0 = nn.Parameter(torch.empty(3 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih_l0 = nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))
        self.bias_hh_l0 = nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))


class GCN_GRU(nn.Module):
    """forward(adj_matrix [S,S], attr_matrix [1,T,S,13]) -> [T, gru_hidden_dim], exactly the
    reference's contract.  Extensions: attr_matrix [B,T,S,13] with B > 1 returns [B,T,H] (B independent windows,
    h0 = 0 each); a float16 / bfloat16 attr_matrix (math "f16x3" or "f16" only) is read as such and the output comes
    back in the same type (16-bit I/O, BASELINE's 16-bit configuration) -- parameters and gradients stay fp32."""

    def __init__(self, input_dim, hidden_dim, output_dim, gru_input, gru_hidden_dim, math="f32", validate=True):
        super().__init__()
        # validate: in the fp16-plane modes read the library's range-status word after every forward (one 4-byte
        # device-to-host copy) and raise instead of returning inf/NaN-derived values; TrainStep checks every N steps
        self.validate = validate
        if not (input_dim == hidden_dim == output_dim == NUM_FEATURES):
            raise RuntimeError("GCN_GRU: the reference hard-codes 13 features per station "
                               "(src/step6_gcn_gru_combined_model.py:16); got %s"
                               % ((input_dim, hidden_dim, output_dim),))
        self.conv1 = GraphConvLayer(input_dim, hidden_dim)
        self.conv2 = GraphConvLayer(hidden_dim, output_dim)
        self.gru = _GRUParams(gru_input, gru_hidden_dim)
        self.math = {"f32": _lib.MATH_F32, "f16x3": _lib.MATH_F16X3, "f16": _lib.MATH_F16, "f16x3g": _lib.MATH_F16X3G}[math]

    def hot_path_parameters(self):
        return (self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                self.gru.weight_ih_l0, self.gru.weight_hh_l0, self.gru.bias_ih_l0, self.gru.bias_hh_l0)

    def forward(self, adj_matrix, attr_matrix):
        if attr_matrix.dim() != 4:
            raise RuntimeError("GCN_GRU.forward: attr_matrix must be [B, T, S, 13], got %s" % (tuple(attr_matrix.shape),))
        B, T, S, F = attr_matrix.shape
        flat = S * NUM_FEATURES
        if F != NUM_FEATURES or flat != self.gru.input_size:
            # mirrors the reference's .view(1, num_seq, flat) failure (step6:20)
            raise RuntimeError("shape '[%d, %d, %d]' is invalid for input of size %d"
                               % (B, T, self.gru.input_size, attr_matrix.numel()))
        out = gcn_gru(adj_matrix, attr_matrix, self.hot_path_parameters(), self.math)
        if self.validate and self.math != _lib.MATH_F32:
            check_range_status(out.device)
        return out.squeeze(0)            # step6:26
