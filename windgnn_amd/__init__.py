"""windgnn_amd: MI355X-native (gfx950 HIP) implementation of WindGNN's GCN+GRU hot path behind
the reference's own nn.Module API.  See DESIGN.md / INTEGRATION.md."""
from .modules import GCN_GRU, GraphConvLayer  # noqa: F401

__all__ = ["GCN_GRU", "GraphConvLayer"]
