"""configs[4] (S = 4096 CSR, H = 12288, B = 128, T = 24) with the large-shape NT plane GEMM on / off (WGNN_OPT_BIG_GEMM), same
process, same box: ms per step and the per-kernel times.    python tools/exp/c5_ab.py [f16x3|f16x3g ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import secondary_c5
from windgnn_amd import _lib
dev = torch.device("cuda:0")
for math in (sys.argv[1:] or ["f16x3", "f16x3g"]):
    for big in (1, 0, 1):
        _lib.set_option(_lib.OPT_BIG_GEMM, big)
        _lib.profile_enable(True)
        r = secondary_c5(dev, math, nsteps=3)
        torch.cuda.synchronize()
        recs = sorted(_lib.profile_read(), key=lambda x: -x["ms"])
        _lib.profile_enable(False)
        ks = "  ".join("%s=%.1fms/%d" % (k["name"].replace("_kernel", ""), k["ms"] / 5, k["launches"] // 5) for k in recs[:8])
        print("c5 %s big_gemm=%d: %.2f ms/step (%.1f windows/s, roofline frac %.4f)   per step [%s]"
              % (math, big, r["ms_per_step"], r["value"], r["roofline"]["frac"], ks), flush=True)
_lib.set_option(_lib.OPT_BIG_GEMM, 1)
