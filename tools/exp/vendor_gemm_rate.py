"""What the vendor GEMM libraries (rocBLAS / hipBLASLt through torch.matmul) reach on this box for the shapes of this path's
products -- a yardstick for the hand-written kernels, not a code path of the library (windgnn_amd never calls them).
  configs[4] (4096 stations):  GI  [3072 x 53248] . [36864 x 53248]^T   one fp16 pass (our split-fp16 product issues three)
                               dg  [3072 x 36864] . [36864 x 53248]
                               dW  [3072 x 36864]^T . [3072 x 53248]
  headline (34 stations):      GI  [98304 x 448] . [320 x 448]^T, dg [98304 x 320] . [320 x 448], dW [98304 x 320]^T [98304 x 448]  fp16 and fp32
    python tools/exp/vendor_gemm_rate.py"""
import torch
dev = torch.device("cuda:0")


def rate(name, a, b, ta, tb, n=10):
    A = a.t() if ta else a
    B = b.t() if tb else b
    for _ in range(3):
        C = A @ B
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        C = A @ B
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    M, K = A.shape
    N = B.shape[1]
    print("%-34s %-8s M=%6d N=%6d K=%6d  %9.3f ms  %7.1f TFLOP/s" % (name, str(a.dtype).replace("torch.", ""), M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    del C


for dt in (torch.float16, torch.bfloat16):
    g = torch.randn(3072, 53248, device=dev, dtype=dt)
    W = torch.randn(36864, 53248, device=dev, dtype=dt) * 0.01
    dGI = torch.randn(3072, 36864, device=dev, dtype=dt)
    rate("c5 GI = g W^T", g, W, False, True)
    rate("c5 dg = dGI W", dGI, W, False, False)
    rate("c5 dW = dGI^T g", dGI, g, True, False)
    del g, W, dGI
    torch.cuda.empty_cache()
for dt in (torch.float16, torch.float32):
    g = torch.randn(98304, 448, device=dev, dtype=dt)
    W = torch.randn(320, 448, device=dev, dtype=dt)
    dGI = torch.randn(98304, 320, device=dev, dtype=dt)
    rate("headline GI = g W^T", g, W, False, True, 30)
    rate("headline dg = dGI W", dGI, W, False, False, 30)
    rate("headline dW = dGI^T g", dGI, g, True, False, 30)
