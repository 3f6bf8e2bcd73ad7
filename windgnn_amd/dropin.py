"""Star-import shim with the same public names as the reference's `step6_gcn_gru_combined_model`
(src/step6_gcn_gru_combined_model.py:1-3 imports `torch`, `torch.nn as nn` and, through
`from step5_gcn_layer_model import *`, `GraphConvLayer`; src/main.py:8 star-imports all of it and then uses
`nn.MSELoss()` at :49 and `torch.optim.Adam` at :52).  Replacing

    from step6_gcn_gru_combined_model import *
with
    from windgnn_amd.dropin import *

leaves every name src/main.py uses from that module in scope."""
import torch                       # noqa: F401
import torch.nn as nn              # noqa: F401

from .modules import GCN_GRU, GraphConvLayer   # noqa: F401

__all__ = ["torch", "nn", "GCN_GRU", "GraphConvLayer"]
