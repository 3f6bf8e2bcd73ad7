"""configs[4] (S = 4096 CSR, H = 12288, B = 128, T = 24): ms per step and EVERY kernel's share (event-bracketed inside the library),
strict and mixed.    python tools/exp/c5_kernels.py [f16x3|f16x3g ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import secondary_c5
from windgnn_amd import _lib
dev = torch.device("cuda:0")
for math in (sys.argv[1:] or ["f16x3", "f16x3g"]):
    r0 = secondary_c5(dev, math, nsteps=3)                       # untimed by the profiler: the step as bench.py reports it
    _lib.profile_enable(True)
    r = secondary_c5(dev, math, nsteps=3)
    torch.cuda.synchronize()
    recs = sorted(_lib.profile_read(), key=lambda x: -x["ms"])
    _lib.profile_enable(False)
    nst = 5                                                      # secondary_c5: 2 warm-up + 3 timed steps
    print("c5 %s: %.2f ms/step (%.1f windows/s); with per-kernel events %.2f ms/step" % (math, r0["ms_per_step"], r0["value"], r["ms_per_step"]), flush=True)
    tot = sum(k["ms"] for k in recs) / nst
    for k in recs:
        print("    %-34s %8.3f ms/step in %5.1f launches (%6.1f us each)  %5.1f %%" % (k["name"].replace("_kernel", ""), k["ms"] / nst, k["launches"] / nst,
              1e3 * k["ms"] / k["launches"], 100.0 * k["ms"] / nst / tot), flush=True)
    print("    sum %.2f ms/step" % tot, flush=True)
