#!/usr/bin/env python3
"""Headline benchmark: windows/s of the WindGNN training step (forward + MSE + backward + Adam, with a
gradient all-reduce when N > 1) on synthetic 34-station hourly windows.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").  Inputs are resident in HBM before the
timed region; every arithmetic op of the step is a libwindgnn_hip.so kernel."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S, T, F, H = 34, 24, 13, 102
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "f16": 2500.0}
# which roofline bounds each kernel (DESIGN.md "Kernels")
BOUND_HBM_PREFIXES = ("mse_", "adam_", "amax_", "splitk_reduce", "tn_reduce", "split_weight", "gcn_partial", "csr_", "gru_cell")


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/pmc_traffic.py), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            k = json.load(open(path))["kernels"].get(kernel)
        except Exception:
            k = None
        if k:
            return {"bytes_per_launch": round(k["bytes_per_launch"]), "source": os.path.basename(path)}
    return None


def adjacency_34():
    z = np.load(os.path.join(ROOT, "tests", "golden", "graph_7_34.npz"))
    return torch.from_numpy(z["A34"]).float()          # src/main.py:26: float64 -> .float()


def make_inputs(B, rank, dev, S=S, H=H):
    g = torch.Generator().manual_seed(1234 + rank)
    X = torch.rand(B, T, S, F, generator=g)
    L = torch.rand(B, T, H, generator=g)
    return X.to(dev), L.to(dev)


def adjacency_knn(S, k=8):
    """BASELINE configs[4] / SURVEY 8d: S points over the station list's span, symmetric k-NN, CSR, seed 7."""
    from windgnn_amd.graph import CsrAdjacency, build_knn_adjacency, synthetic_station_coords
    return CsrAdjacency(*build_knn_adjacency(synthetic_station_coords(S, seed=7), k))


def host_threads():
    """Threads this process may really use: affinity mask, cgroup cpu.max quota, and the GPU box's
    16-core share per GPU (os.cpu_count() reports the whole host)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(budget_s=12.0, Bc=1024):
    """The oracle (a torch-CPU restatement of the reference ops) timed on this host: same step
    (forward + MSE + hand-derived backward + Adam), fp32, batched, all host threads."""
    from oracle import windgnn_oracle as orc
    torch.set_num_threads(host_threads())
    A = adjacency_34()
    g = torch.Generator().manual_seed(1234)
    X = torch.rand(Bc, T, S, F, generator=g)
    L = torch.rand(Bc, T, H, generator=g)
    p = orc.init_params(S, F, H, seed=0)
    st = orc.adam_init(p)
    _, _, gr = orc.train_step(A, X, L, p)              # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        _, _, gr = orc.train_step(A, X, L, p)
        p = orc.adam_step(p, gr, st)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 50:
            break
    return {"value": round(n * Bc / dt, 1), "unit": "windows/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps of B=%d windows (S=34,T=24,H=102 fp32, fwd+MSE+bwd+Adam), oracle/windgnn_oracle.py "
                      "on torch %s CPU, %.1f s" % (n, Bc, torch.__version__, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="windows per GPU (weak scaling); default 4096 (128 for c5)")
    ap.add_argument("--math", default="f16x3", choices=["f32", "f16x3", "f16"])
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"],
                    help="c3: 34 stations, the headline config; c5: 4096-station k-NN CSR stress config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d != WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from windgnn_amd import GCN_GRU, _lib
    from windgnn_amd.trainer import TrainStep

    torch.manual_seed(0)                                # identical parameters on every rank
    S, H = (34, 102) if args.workload == "c3" else (4096, 12288)
    B = args.batch or (4096 if args.workload == "c3" else 128)
    model = GCN_GRU(F, F, F, S * F, H, math=args.math).to(dev)
    trainer = TrainStep(model)
    A = (adjacency_34() if args.workload == "c3" else adjacency_knn(S)).to(dev)
    X, L = make_inputs(B, rank, dev, S, H)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(A, X, L)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = trainer.step(A, X, L)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- per-kernel pass (same workload, hipEvents inside the library) for the roofline object
    roofline, forward, kernels, mfma = None, None, None, None
    if rank == 0:
        _lib.profile_enable(True)
        for _ in range(args.steps):
            trainer.forward_backward(A, X, L)
        torch.cuda.synchronize()
        recs = _lib.profile_read()
        _lib.profile_enable(False)
        recs.sort(key=lambda r: -r["ms"])
        kernels = [{"name": r["name"], "launches_per_step": r["launches"] / args.steps,
                    "avg_us": round(1e3 * r["ms"] / max(r["launches"], 1), 2),
                    "ms_per_step": round(r["ms"] / args.steps, 4)} for r in recs]
        d = recs[0]
        avg_s = d["ms"] / d["launches"] * 1e-3
        # which roofline binds this kernel: time its ALGORITHMIC bytes need at the HBM peak vs the time its
        # algorithmic flops need at the MFMA peak (f16x3 issues 3 MFMA passes per product => peak / 3)
        eff_peak = MFMA_PEAK_TFLOPS[args.math] / (3.0 if args.math == "f16x3" else 1.0)
        t_hbm = d["bytes"] / (HBM_PEAK_GBS * 1e9)
        t_mfma = d["flops"] / (eff_peak * 1e12)
        bound = "hbm" if (d["name"].startswith(BOUND_HBM_PREFIXES) or t_hbm >= t_mfma) else "mfma"
        tr = measured_traffic(d["name"]) if args.workload == "c3" else None   # the PMC passes are of the c3 workload
        if bound == "mfma":
            ach = d["flops"] / d["launches"] / avg_s / 1e12
            peak = MFMA_PEAK_TFLOPS[args.math]
            roofline = {"kernel": d["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": tr["bytes_per_launch"] if tr else None,
                        "avg_launch_us": round(avg_s * 1e6, 2)}
        else:
            ach = d["bytes"] / d["launches"] / avg_s / 1e9
            roofline = {"kernel": d["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": tr["bytes_per_launch"] if tr else None,
                        "avg_launch_us": round(avg_s * 1e6, 2)}
        # MFMA side of the picture (north-star: "HBM GB/s and MFMA utilisation against the roofline"): the GEMM with
        # the most time; f16x3 issues 3 fp16 MFMA passes per algorithmic FLOP
        gemms = [r for r in recs if r["name"].startswith(("pgemm_", "gemm_f32")) and r["flops"] > 0]
        mfma = None
        if gemms:
            gk = gemms[0]
            passes = 3.0 if (args.math == "f16x3" and gk["name"].startswith("pgemm_")) else 1.0
            peak_k = MFMA_PEAK_TFLOPS["f32" if gk["name"].startswith("gemm_f32") else args.math]
            tf = gk["flops"] / (gk["ms"] * 1e-3) / 1e12
            mfma = {"kernel": gk["name"], "algorithmic_TFLOPs": round(tf, 1), "issued_TFLOPs": round(tf * passes, 1),
                    "peak": peak_k, "frac_issued": round(tf * passes / peak_k, 4)}
        roofline["algorithmic_bytes_per_launch"] = round(d["bytes"] / d["launches"])
        roofline["hbm_GBs_algorithmic"] = round(d["bytes"] / d["launches"] / avg_s / 1e9, 1)
        if tr:
            roofline["traffic_source"] = tr["source"]
        # forward-only timing: north-star "fused forward vs HBM roofline" (52 224 algorithmic B/window)
        from windgnn_amd.functional import gcn_gru_forward_raw
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            gcn_gru_forward_raw(A, X, trainer.p_views, model.math, want_stash=False)
        e1.record()
        torch.cuda.synchronize()
        fwd_s = e0.elapsed_time(e1) * 1e-3 / args.steps
        fwd_bytes = B * T * (S * F + H) * 4.0
        forward = {"us": round(fwd_s * 1e6, 1), "algorithmic_GBs": round(fwd_bytes / fwd_s / 1e9, 1),
                   "hbm_frac": round(fwd_bytes / fwd_s / 1e9 / HBM_PEAK_GBS, 4),
                   "windows_per_s": round(B / fwd_s, 1)}

    if rank == 0:
        out = {
            "metric": "windows/sec (fwd+bwd) %s seq24 GCN-GRU" % ("34-node" if args.workload == "c3" else "4096-node"),
            "value": round(world * B * args.steps / dt, 1),
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f16x3": "f16x3(split-fp32)", "f16": "f16"}[args.math],
            "data": "synthetic",
            "config": {"workload": "S=%d stations%s, T=24, F=13, H=%d, B=%d windows/GPU; step = forward + MSE + "
                                   "backward + grad all-reduce (N>1) + Adam; fp32 I/O"
                                   % (S, "" if args.workload == "c3" else " (symmetric 8-NN graph, CSR)", H, B),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "math": args.math},
            "loss": round(float(loss), 6),
            "roofline": roofline,
            "mfma": mfma,
            "forward": forward,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "c3":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
