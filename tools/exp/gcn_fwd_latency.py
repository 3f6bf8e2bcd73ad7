"""r4 probe: per-tile latency of gcnx_fwd with a lighter memory skeleton / fewer resident waves (what a fused GCN -> GI kernel
would give its GCN waves).  The knobs (WGNN_EXP_FWD_GRID / _LDS / _STORE) existed in launch_gcnx2_fwd only in the experiment
build of commit "gcngi.hip: fused GCN -> input-projection forward" (r4); the results are profiles/r4_probe_gcn_fwd_latency.txt.
    MATH=f16x3|f16 python tools/exp/gcn_fwd_latency.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.functional import gcn_gru_forward_raw
dev = torch.device("cuda:0")
math = os.environ.get("MATH", "f16x3")
m = GCN_GRU(F, F, F, S * F, H, math=math).to(dev)
A = adjacency_34().to(dev)
X, L = make_inputs(4096, 0, dev)
params = [q.detach() for q in m.hot_path_parameters()]
# dirty the Infinity Cache between calls like the training step does: a 1 GB write
junk = torch.empty(256 << 20, dtype=torch.float32, device=dev)


def run(tag, grid=None, lds=None, store=None, dirty=True):
    for k, v in (("WGNN_EXP_FWD_GRID", grid), ("WGNN_EXP_FWD_LDS", lds), ("WGNN_EXP_FWD_STORE", store)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    for _ in range(3):
        gcn_gru_forward_raw(A, X, params, m.math, want_stash=False)
    _lib.profile_enable(True)
    n = 10
    for _ in range(n):
        if dirty:
            junk.fill_(1.0)
        gcn_gru_forward_raw(A, X, params, m.math, want_stash=False)
    torch.cuda.synchronize()
    recs = _lib.profile_read()
    _lib.profile_enable(False)
    t = [1e3 * r["ms"] / r["launches"] for r in recs if r["name"].startswith("gcnx_fwd")][0]
    print("%-58s gcnx_fwd = %6.1f us" % (tag, t), flush=True)


for dirty in (True, False):
    print("dirty cache in front of every call: %s" % dirty)
    run("baseline: 512 blocks x 8 waves (16 waves/CU), both planes", dirty=dirty)
    run("hi plane only", store=1, dirty=dirty)
    run("no stores", store=2, dirty=dirty)
    run("8 waves/CU (256 blocks, 1 per CU), both planes", grid=256, lds=90000, dirty=dirty)
    run("8 waves/CU, hi only", grid=256, lds=90000, store=1, dirty=dirty)
    run("8 waves/CU, no stores", grid=256, lds=90000, store=2, dirty=dirty)
    run("16 waves/CU as 512 blocks, no stores (check)", grid=512, store=2, dirty=dirty)
