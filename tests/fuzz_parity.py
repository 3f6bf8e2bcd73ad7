"""Randomised parity run (not part of the test suite: a longer, wider version of test_random_shape_sweep / test_random_csr_shape_sweep
for the GPU box): N random shapes per math mode through the drop-in module -- dense and CSR adjacency, training step and
stash-less forward -- against the fp64 oracle at the mode's tolerance.  Prints every violation and a summary.
H >= 2: with ONE hidden unit every gate of every window hangs on the same three weights rows, the fp32 CPU evaluation itself is
2e-5 ... 2e-2 off fp64 and the split-fp16 modes another factor 3-5 beyond it (2.8e-4 at S = 16, B = 56: profiles/r5_fuzz_parity.txt).
    python tests/fuzz_parity.py [--cases 200] [--seed 1] [--modes f32,f16x3,f16x3g,f16]"""
import argparse, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import windgnn_oracle as orc
from windgnn_amd.graph import CsrAdjacency, build_knn_adjacency, synthetic_station_coords
from test_gpu_parity import _model_from, _run_step, PARAM_KEYS, rel_to_max, max_abs, Y_TOL, G_TOL, F16_Y_TOL, F16_G_TOL

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--modes", default="f32,f16x3,f16x3g,f16")
args = ap.parse_args()
dev = torch.device("cuda:0")
t00 = time.time()
total_bad = 0
for math in args.modes.split(","):
    rnd = random.Random(args.seed * 1000 + len(math) * 7 + ord(math[-1]))
    y_tol, g_tol = (F16_Y_TOL, F16_G_TOL) if math == "f16" else (Y_TOL, G_TOL)
    bad, worst_y, worst_g, ncsr, nsat, nbound, nrelaxed = [], 0.0, 0.0, 0, 0, 0, 0
    for case in range(args.cases):
        csr_case = rnd.random() < 0.3
        if csr_case:
            S = rnd.choice([rnd.randint(2, 64), rnd.randint(65, 400), rnd.randint(401, 2300)])
            T, B, H = rnd.randint(1, 4), rnd.randint(1, 6), rnd.randint(2, 200 if rnd.random() < 0.15 else 120)
        else:
            S = rnd.randint(1, 64)
            T, B = rnd.randint(1, 30), rnd.randint(1, 70)
            H = rnd.randint(2, 200 if rnd.random() < 0.1 else 128)
            if rnd.random() < 0.1:
                B = rnd.randint(171, 400)                       # B * T beyond 4096 rows: the large-batch kernels
                T = rnd.randint(12, 24)
        g = torch.Generator().manual_seed(args.seed * 100000 + case)
        if csr_case:
            k = rnd.randint(1, min(12, max(S - 1, 1)))
            if S < 2:
                continue
            adj = CsrAdjacency(*build_knn_adjacency(synthetic_station_coords(S, seed=case), k))
            A = adj.dense()
            ncsr += 1
        else:
            k = 0
            A = torch.rand(S, S, generator=g) / S + 0.01
            adj = A
        X = torch.rand(B, T, S, 13, generator=g) * min(1.0, 34.0 / S)     # (keeps the gates out of saturation for big graphs)
        L = torch.rand(B, T, H, generator=g)
        p = orc.init_params(S, 13, H, seed=case)
        pd = {k2: v.double() for k2, v in p.items()}
        Yo, loss_o, go = orc.train_step(A.double(), X.double(), L.double(), pd)
        # two ways a well-posed kernel can still disagree with the fp64 oracle, both properties of the PROBLEM, reported apart:
        #  * a ReLU pre-activation within fp32 rounding of zero: its mask -- and one gradient contribution -- flips with the
        #    summation order (1 case in ~15 at these sizes: |Z| = 3e-8 seen at S = 26, B = 52);
        #  * saturated gates (a few hidden units over many inputs): every gradient is below 1e-9 and cancels in fp32.
        H1o, P1o = orc.gcn_layer_fwd(A.double(), X.double(), pd["conv1.weight"], pd["conv1.bias"])
        Z1 = torch.matmul(P1o, pd["conv1.weight"]) + pd["conv1.bias"]
        H2o, P2o = orc.gcn_layer_fwd(A.double(), H1o, pd["conv2.weight"], pd["conv2.bias"])
        Z2 = torch.matmul(P2o, pd["conv2.weight"]) + pd["conv2.bias"]
        zmin = min(float(Z1.abs().min() / Z1.abs().max().clamp_min(1e-30)), float(Z2.abs().min() / Z2.abs().max().clamp_min(1e-30)))
        boundary = zmin < 3e-7                                 # fp32 rounding of a 13 ... 64-term sum of O(scale) terms
        gmax = max(float(v.abs().max()) for v in go.values())
        #  * a handful of hidden units over hundreds of inputs (H <= 6 or so): the gates sit far out on their sigmoids and the
        #    plain fp32 CPU evaluation of the same formulas -- what the reference itself computes -- is 1e-4 ... 4e-4 off the fp64
        #    result.  The bar of such a case is 4x that fp32 evaluation's own error where it exceeds the mode's tolerance.
        _, _, g32 = orc.train_step(A, X, L, p)
        cond = {key: rel_to_max(g32[key], go[key]) for key in PARAM_KEYS}
        tag = (math, case, "csr k=%d" % k if csr_case else "dense", S, T, B, H) + (("relu-boundary %.0e" % zmin,) if boundary else ())
        try:
            model = _model_from(p, S, H, math)
            a_dev = adj.to(dev)
            out, loss, grads = _run_step(model, a_dev, X.to(dev), L.to(dev))
            with torch.no_grad():
                out2 = model(a_dev, X.to(dev)).cpu()
            # the same step through TrainStep (wgnn_fwd_loss, deferred backward parts, wgnn_finish with Adam): its loss and the
            # gradients it leaves in its bucket are those of the parameters BEFORE the update
            from windgnn_amd.trainer import TrainStep
            tr = TrainStep(_model_from(p, S, H, math))
            loss_t, _ = tr.step(a_dev, X.to(dev), L.to(dev))
            tr.check()
            grads_t = {key: t.detach().cpu().clone() for key, t in zip(PARAM_KEYS, tr.g_views)}
            tr.close()
        except Exception as e:                                  # a refused shape is a finding too
            bad.append((tag, "raised %s: %s" % (type(e).__name__, str(e)[:120])))
            continue
        ey = max(max_abs(out.reshape(Yo.shape), Yo), max_abs(out2.reshape(Yo.shape), Yo))
        worst_y = max(worst_y, ey)
        if ey > y_tol:
            bad.append((tag, "Y %.2e" % ey))
        if abs(float(loss_t) - float(loss_o)) > (1e-5 if math != "f16" else 2e-2) * max(1.0, float(loss_o)):
            bad.append((tag, "TrainStep loss %.6f vs %.6f" % (float(loss_t), float(loss_o))))
        if gmax < 1e-8:                                         # saturated: nothing to compare against
            nsat += 1
            continue
        for key in PARAM_KEYS:
            if boundary and key.startswith("conv"):
                continue
            if float(go[key].abs().max()) < 1e-9 * gmax:       # this tensor's gradient is noise next to the others (dW_hh at T = 1 ...)
                continue
            e = rel_to_max(grads[key], go[key])
            bar = max(g_tol, 4.0 * cond[key])
            if bar > g_tol:
                nrelaxed += 1
            else:
                worst_g = max(worst_g, e)
            if e > bar:
                bad.append((tag, "%s %.2e (bar %.1e; fp32 CPU evaluation: %.1e)" % (key, e, bar, cond[key])))
            et = rel_to_max(grads_t[key], go[key])
            if et > bar:
                bad.append((tag, "TrainStep %s %.2e (bar %.1e; module path %.2e)" % (key, et, bar, e)))
        nbound += int(boundary)
    print("%-7s %d cases (%d CSR; %d with a ReLU pre-activation within fp32 rounding of 0: conv gradients not compared; %d saturated: no gradient compared): "
          "worst |Y - oracle| %.2e (bar %.0e), worst gradient error / max %.2e (bar %.0e; %d tensors of ill-conditioned cases held to 4x the fp32 CPU evaluation's own error instead), %d violations"
          % (math, args.cases, ncsr, nbound, nsat, worst_y, y_tol, worst_g, g_tol, nrelaxed, len(bad)), flush=True)
    for b in bad:
        print("    ", b, flush=True)
    total_bad += len(bad)
print("done in %.0f s, %d violations" % (time.time() - t00, total_bad), flush=True)
