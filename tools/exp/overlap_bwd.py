"""Backward parts on two streams: after the BPTT recurrence (part 1) the weight-gradient GEMMs (part 4: pgemm_tn x2, bound by
HBM) and the dg GEMM + GCN backward (part 2: the GCN backward is bound by its SIMDs' issue rate) are independent.
    python tools/exp/overlap_bwd.py [steps] [modes: seq two "dg|tn"]
Measured (r4, B = 4096, f16x3, us per step, two rounds each): seq 750 / 748, two 764 / 762 (the persistent GCN backward takes every
CU, nothing overlaps, the two stream hops cost ~14 us); with the GCN backward capped to 128 workgroups (experiment build) seq
856 / 855, two 780 / 775 (the TN GEMMs do run beside it, but 128 CUs are too few for it); "dg|tn" (experiment build that
splits part 2: dg GEMM beside the TN GEMMs, GCN backward after both) seq 740 / 735 vs 760 / 757.  Kernels that each need a
whole CU (LDS) do not share CUs, and the hop cost is real: negative."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.functional import finish_step, gcn_gru_backward_mse_raw
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = GCN_GRU(F, F, F, S * F, H, math=os.environ.get("MATH", "f16x3")).to(dev)
tr = TrainStep(m)
A = adjacency_34().to(dev)
X, L = make_inputs(4096, 0, dev)
tr.step(A, X, L)
DEFER = _lib.BWD_DEFER
side = torch.cuda.Stream()
e1, e2 = torch.cuda.Event(), torch.cuda.Event()


def step(mode):
    Y, stash, d = tr._forward(A, X, L)
    pre, loss = tr._prepared, tr._loss
    bw = lambda part: gcn_gru_backward_mse_raw(d, A, X, tr.p_views, Y, L, stash, tr.g_views, loss, 1.0, part=part | DEFER, prepared=pre)
    bw(1 | 8)
    if mode == "seq":
        bw(2); bw(4)
    elif mode == "dg|tn":          # dg GEMM on the main stream next to the two TN GEMMs on the side stream, the GCN backward after both
        # the committed library never read WGNN_EXP_PART2 (the split of part 2 lived in an experiment build that was
        # not kept), so on the shipped tree this mode would run part 2 twice and time nonsense.  Refused unless such a build
        # announces itself.
        if os.environ.get("WGNN_LIB_SPLITS_PART2") != "1":
            raise SystemExit('overlap_bwd.py: mode "dg|tn" needs the round-4 experiment build that splits backward part 2 '
                             "(WGNN_EXP_PART2); the committed library does not, so its numbers in the docstring cannot be "
                             "reproduced from this tree.  Modes seq / two run on the shipped library.")
        main = torch.cuda.current_stream()
        e1.record(main)
        side.wait_event(e1)
        with torch.cuda.stream(side):
            bw(4)
            e2.record(side)
        os.environ["WGNN_EXP_PART2"] = "1"
        bw(2)
        main.wait_event(e2)
        os.environ["WGNN_EXP_PART2"] = "2"
        bw(2)
        os.environ["WGNN_EXP_PART2"] = "0"
    else:
        main = torch.cuda.current_stream()
        e1.record(main)
        side.wait_event(e1)
        with torch.cuda.stream(side):
            bw(4)
            e2.record(side)
        bw(2)
        main.wait_event(e2)
    finish_step(d, tr.p_views, tr.g_views, 6, tr._adam(), pre, tr.device)
    tr.steps += 1


for mode in (sys.argv[2:] or ["seq", "two", "seq", "two"]):
    for _ in range(20):
        step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(mode)
    torch.cuda.synchronize()
    print("%s: %.1f us per step, loss %.6f" % (mode, (time.perf_counter() - t0) / n * 1e6, float(tr._loss)), flush=True)
