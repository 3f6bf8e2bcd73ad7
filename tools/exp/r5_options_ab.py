"""Round-5 same-process A/Bs through wgnn_set_option (one library, one box, one process):
  (a) the fused front end's wave roles per SIMD (WGNN_OPT_GG_ROLE_SPLIT) and s_setprio on its GEMM waves (WGNN_OPT_GG_GEMM_PRIO):
      stash-less forward at B = 4096, f16x3 and one-pass f16 + bf16 I/O;
  (b) backward part 2 as producer -> consumer pairs over row chunks (WGNN_OPT_BWD2_CHUNKS = 1 / 2 / 4 / 8): the training step,
      strict f16x3 and f16x3g.
    python tools/exp/r5_options_ab.py [a] [b]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.functional import gcn_gru_forward_raw
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
what = sys.argv[1:] or ["a", "b"]
A = adjacency_34().to(dev)


def timed(fn, n=40, warm=10):
    for _ in range(warm):
        fn()
    _lib.profile_enable(True)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    recs = _lib.profile_read()
    _lib.profile_enable(False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, "  ".join("%s=%.1f" % (r["name"].replace("_kernel", ""), 1e3 * r["ms"] / r["launches"]) for r in recs)


if "a" in what:
    for math, io in (("f16x3", "fp32"), ("f16", "bf16")):
        m = GCN_GRU(F, F, F, S * F, H, math=math).to(dev)
        tr = TrainStep(m)
        X, L = make_inputs(4096, 0, dev, io=io)
        tr.step(A, X, L)
        fwd = lambda: gcn_gru_forward_raw(A, X, tr.p_views, m.math, want_stash=False, prepared=tr._prepared)
        ref = fwd()[0].clone()
        for rnd in range(2):
            for split, prio in ((0, 0), (1, 0), (0, 1), (0, 2), (0, 3), (1, 1), (0, 0)):
                _lib.set_option(_lib.OPT_GG_ROLE_SPLIT, split)
                _lib.set_option(_lib.OPT_GG_GEMM_PRIO, prio)
                same = torch.equal(fwd()[0], ref)
                us, ks = timed(fwd)
                print("(a) %s io=%s round %d  role_split=%d gemm_prio=%d  bitwise=%s  forward %.1f us  [%s]" % (math, io, rnd, split, prio, same, us, ks), flush=True)
        _lib.set_option(_lib.OPT_GG_ROLE_SPLIT, 0)
        _lib.set_option(_lib.OPT_GG_GEMM_PRIO, 0)
        del tr, m

if "b" in what:
    for math in ("f16x3", "f16x3g"):
        m = GCN_GRU(F, F, F, S * F, H, math=math).to(dev)
        tr = TrainStep(m)
        X, L = make_inputs(4096, 0, dev)
        step = lambda: tr.step(A, X, L)
        for rnd in range(2):
            for ch in (1, 2, 4, 8, 1):
                _lib.set_option(_lib.OPT_BWD2_CHUNKS, ch)
                us, ks = timed(step)
                print("(b) %s round %d  bwd2_chunks=%d  step %.1f us  [%s]" % (math, rnd, ch, us, ks), flush=True)
        _lib.set_option(_lib.OPT_BWD2_CHUNKS, 1)
        del tr, m
