#!/usr/bin/env python3
"""Headline benchmark: windows/s of the WindGNN training step (forward + MSE + backward + Adam, with a
gradient all-reduce when N > 1) on synthetic 34-station hourly windows.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").  Inputs are resident in HBM before the
timed region; every arithmetic op of the step is a libwindgnn_hip.so kernel.

Order of a default N = 1 run: (1) two short child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE, then
WRITE_SIZE: the counters cannot share a pass) measure the HBM traffic of every kernel of THIS build, before this
process touches the GPU; (2) a per-kernel pass with hipEvents inside the library and the forward-only timing (so the
GPU does not enter the timed region cold); (3) `warmup` untimed steps + exactly `steps` timed steps; (4) the exact-fp32
mode on the same workload (secondary value); (5) the CPU baselines."""
import argparse
import json
import re
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S, T, F, H = 34, 24, 13, 102
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "f16x3g": 2500.0, "f16": 2500.0}
DTYPE_LABEL = {"f32": "f32", "f16x3": "f16x3(split-fp32)", "f16": "f16",
               "f16x3g": "f16x3g(split-fp32 forward + recurrences; gate gradients as one fp16 plane in the backward GEMMs)"}
# WGNN_MATH_F16X3G's measured gradient error (max |g - g_fp64| / max |g_fp64| over the 8 tensors), carried next to its number
F16X3G_GRAD_ERR = {"mse_step_B4096": 2.0e-6, "strict_f16x3_same_test": 9.6e-7, "zero_mean_noise_dY_B4096": 3.7e-4, "bar": 1e-4,
                   "source": "tests/test_gpu_parity.py::test_full_size_B4096_mse_gradients_against_fp64_oracle, "
                             "::test_full_size_properties_B4096[f16x3g]"}
# which roofline bounds each kernel (DESIGN.md "Kernels")
BOUND_HBM_PREFIXES = ("mse_", "adam_", "finish_", "amax_", "splitk_reduce", "tn_reduce", "split_weight", "gcn_partial", "csr_", "gru_cell")


SPINUP_STEPS = int(os.environ.get("WGNN_BENCH_SPINUP", "60"))   # untimed steps in front of the W warm-up steps (the GPU's clock ramp after an idle gap: see main())


def committed_traffic():
    """Fallback only: the newest committed PMC summary of the TRAINING STEP (profiles/rN_x_traffic.json; the forward-only
    summaries `*_fwd_only_*_traffic.json` hold other launch counts), or ({}, None).  Newest = highest round tag, not mtime
    (a checkout gives every file the same time)."""
    import glob
    paths = [q for q in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")) if "fwd_only" not in os.path.basename(q)]
    def tag(q):                                          # "r5_b_traffic.json" -> (5, "b")
        m = re.match(r"r(\d+)_([a-z]+)_", os.path.basename(q))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")
    for path in sorted(paths, key=tag, reverse=True):
        try:
            return json.load(open(path))["kernels"], os.path.basename(path)
        except Exception:
            continue
    return {}, None


def measure_traffic_live(args):
    """HBM bytes per launch of every kernel, measured NOW: this script re-run as a child (3 steps) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again with WRITE_SIZE (MI355X_MICROARCH.md, HBM / rocprofv3:
    separate passes; KiB units; gfx950 reads x2).  Returns ({kernel: {...}}, source string) or ({}, reason)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return {}, "rocprofv3 not found"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD"):
        return {}, "already running under a profiler"
    from tools.pmc_traffic import traffic
    tmp = tempfile.mkdtemp(prefix="wgnn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--traffic-child", "--steps", "3", "--warmup", "1",
             "--math", args.math, "--workload", args.workload, "--io", args.io] + (["--batch", str(args.batch)] if args.batch else [])
    try:
        for counter, sub in (("FETCH_SIZE", "F"), ("WRITE_SIZE", "W")):
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d",
                                os.path.join(tmp, sub), "--"] + child, cwd="/tmp", env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=420)
            if r.returncode != 0:
                return {}, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode()[-200:])
        k = traffic(os.path.join(tmp, "F"), os.path.join(tmp, "W"))
        return k, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (reads x2, KiB units)"
    except Exception as e:      # a profiler problem must not cost the bench line
        return {}, "live measurement failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def adjacency_34():
    z = np.load(os.path.join(ROOT, "tests", "golden", "graph_7_34.npz"))
    return torch.from_numpy(z["A34"]).float()          # src/main.py:26: float64 -> .float()


IO_DTYPES = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def make_inputs(B, rank, dev, S=S, H=H, io="fp32"):
    g = torch.Generator().manual_seed(1234 + rank)
    X = torch.rand(B, T, S, F, generator=g)
    L = torch.rand(B, T, H, generator=g)
    return X.to(IO_DTYPES[io]).to(dev), L.to(IO_DTYPES[io]).to(dev)


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(budget_s=10.0, Bc=1024, faithful_budget_s=8.0):
    """The oracle (a torch-CPU restatement of the reference ops) timed on this host, two variants:
    batched   -- one [Bc,T,S,13] call per step: forward + MSE + hand-derived backward + Adam;
    faithful  -- what src/main.py:65-80 does: a loop of reference-shaped B = 1 calls, each followed by its own
                 optimiser step (one window per step)."""
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
    p1 = orc.init_params(S, F, H, seed=0)
    st1 = orc.adam_init(p1)
    for b in range(3):                                  # warm-up
        orc.train_step(A, X[b:b + 1], L[b:b + 1], p1)
    m, t1 = 0, time.perf_counter()
    while True:
        b = m % Bc
        _, _, gr = orc.train_step(A, X[b:b + 1], L[b:b + 1], p1)
        p1 = orc.adam_step(p1, gr, st1)
        m += 1
        dt1 = time.perf_counter() - t1
        if dt1 >= faithful_budget_s:
            break
    shape = "S=34,T=24,H=102 fp32, fwd+MSE+bwd+Adam, oracle/windgnn_oracle.py on torch %s CPU" % torch.__version__
    return {"value": round(n * Bc / dt, 1), "unit": "windows/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(),
            "sample": "batched: %d steps of B=%d windows (%s), %.1f s" % (n, Bc, shape, dt),
            "faithful": {"value": round(m / dt1, 1), "unit": "windows/s",
                         "sample": "%d sequential B=1 steps as src/main.py:65-80 (%s), %.1f s" % (m, shape, dt1)}}


def check_stashless_forward(A, X, L, trainer, math):
    """The forward-only figure times the stash-less forward (the fused GCN + projection kernel where it applies); VERDICT r4
    weak 1: it was timed without its output ever being looked at at this size.  One call of each form on the SAME parameters:
    the stash-less Y must be the training forward's Y bit for bit (the training forward is what the full-size parity tests
    hold against the fp64 oracle).  Raises otherwise: a wrong forward must not produce a bench line."""
    from windgnn_amd.functional import gcn_gru_forward_raw
    y_train = gcn_gru_forward_raw(A, X, trainer.p_views, math, want_stash=True, labels=L, prepared=trainer._prepared)[0]
    y_inf = gcn_gru_forward_raw(A, X, trainer.p_views, math, want_stash=False, prepared=trainer._prepared)[0]
    if not torch.equal(y_train, y_inf):
        raise RuntimeError("bench.py: the stash-less forward's Y differs from the training forward's (max |d| = %.3e)"
                           % float((y_train.float() - y_inf.float()).abs().max()))


def secondary_config(math, io, B, A, dev, nsteps, note, forward=False):
    """One more configuration timed on the same box: `nsteps` full training steps (TrainStep.step) after 50 warm-up steps,
    inputs resident; with `forward` also the forward-only time against its own algorithmic bytes (X + Y in the I/O type)."""
    from windgnn_amd import GCN_GRU
    from windgnn_amd.functional import gcn_gru_forward_raw
    from windgnn_amd.trainer import TrainStep
    torch.manual_seed(0)
    m = GCN_GRU(F, F, F, S * F, H, math=math).to(dev)
    tr = TrainStep(m)
    X, L = make_inputs(B, 0, dev, S, H, io)
    for _ in range(10 + SPINUP_STEPS):                   # first launches of this mode's kernels + the clock ramp (see main())
        tr.step(A, X, L)
    torch.cuda.synchronize()
    s0 = time.perf_counter()
    for _ in range(nsteps):
        tr.step(A, X, L)
    torch.cuda.synchronize()
    dt = time.perf_counter() - s0
    tr.check()
    out = {"dtype": DTYPE_LABEL[math] + ("" if io == "fp32" else " math, %s I/O" % io),
           "value": round(B * nsteps / dt, 1), "unit": "windows/s", "ms_per_step": round(1e3 * dt / nsteps, 4),
           "steps": nsteps, "batch": B, "note": note}
    if forward:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        check_stashless_forward(A, X, L, tr, m.math)
        for _ in range(SPINUP_STEPS):                   # untimed, after the check's host sync: the clock ramp (see main())
            tr.step(A, X, L)
        for _ in range(20):                             # untimed: first launches of the stash-less forward's own kernels
            gcn_gru_forward_raw(A, X, tr.p_views, m.math, want_stash=False, prepared=tr._prepared)
        nfwd = max(nsteps, 50)
        e0.record()
        for _ in range(nfwd):
            gcn_gru_forward_raw(A, X, tr.p_views, m.math, want_stash=False, prepared=tr._prepared)
        e1.record()
        torch.cuda.synchronize()
        fs = e0.elapsed_time(e1) * 1e-3 / nfwd
        esz = 4.0 if io == "fp32" else 2.0
        fb = B * T * (S * F + H) * esz
        out["forward"] = {"us": round(fs * 1e6, 1), "algorithmic_bytes_per_window": int(T * (S * F + H) * esz),
                          "algorithmic_GBs": round(fb / fs / 1e9, 1), "hbm_frac": round(fb / fs / 1e9 / HBM_PEAK_GBS, 4)}
    return out


def secondary_c5(dev, math="f16x3", nsteps=3, B=128):
    """BASELINE configs[4], one GPU's shard: 4096-station symmetric 8-NN graph in CSR, H = 12288, B = 128 windows, T = 24 --
    `nsteps` full training steps after 2 warm-up steps, with the MFMA-side roofline (the step is a dense-GEMM benchmark with a
    negligible SpMM in front: SURVEY section 7).  Parameters are created on the GPU (2.4 G of them)."""
    from windgnn_amd import GCN_GRU
    from windgnn_amd.trainer import TrainStep
    S5, H5 = 4096, 12288
    torch.manual_seed(0)
    with torch.device(dev):
        m = GCN_GRU(F, F, F, S5 * F, H5, math=math)
    tr = TrainStep(m)
    A = adjacency_knn(S5).to(dev)
    X, L = make_inputs(B, 0, dev, S5, H5, "fp32")
    for _ in range(2):
        tr.step(A, X, L)
    torch.cuda.synchronize()
    s0 = time.perf_counter()
    for _ in range(nsteps):
        tr.step(A, X, L)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - s0) / nsteps
    tr.check()
    G3, I5 = 3 * H5, S5 * F
    flops = 6.0 * B * T * G3 * (I5 + H5)                 # GRU products, forward + backward (the GCN layers add < 0.1 %)
    passes = 3.0 if math == "f16x3" else 1.0
    out = {"dtype": DTYPE_LABEL[math] if math != "f16x3g" else
           "f16x3g (mixed: f16x3 forward; backward GEMMs leave the lo plane of the gate gradients unread: dW_ih one MFMA pass, dW_hh / dg two)",
           "value": round(B / dt, 1), "unit": "windows/s", "ms_per_step": round(1e3 * dt, 2), "steps": nsteps, "batch": B,
           "note": "BASELINE configs[4], one GPU's shard: S=4096 stations (symmetric 8-NN graph, CSR), H=12288, T=24, fp32 I/O; "
                   "step = forward + MSE + backward + Adam over 2.42 G parameters",
           "roofline": {"bound": "mfma", "algorithmic_TFLOPs": round(flops / dt / 1e12, 1), "peak": MFMA_PEAK_TFLOPS[math],
                        "unit": "TFLOP/s", "frac": round(flops / dt / 1e12 / MFMA_PEAK_TFLOPS[math], 4),
                        "note": ("every product is %s fp16 MFMA pass(es) with fp32 accumulation: frac counts ALGORITHMIC flops "
                                 "against the dense fp16 peak" % ("3" if passes == 3.0 else "3 (forward) / 1-2 (backward)"))}}
    del tr, m, X, L, A
    torch.cuda.empty_cache()
    return out


def c5_in_child(math, timeout_s=420):
    """secondary_c5(math) in a fresh child process (`bench.py --c5-child MATH` prints its dict as one JSON line)."""
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--c5-child", math], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=timeout_s, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "c5 child (%s) exceeded %d s and was killed" % (math, timeout_s)}
    if r.returncode != 0:
        return {"error": "c5 child (%s) exited with %d: %s" % (math, r.returncode, r.stderr.decode(errors="replace")[-300:])}
    try:
        return json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:
        return {"error": "c5 child (%s): unreadable output: %r" % (math, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="windows per GPU (weak scaling); default 4096 (128 for c5)")
    ap.add_argument("--math", default="f16x3", choices=["f32", "f16x3", "f16x3g", "f16"],
                    help="f16x3 (default, the fp32-parity mode: every product three-pass split fp16, every -m gpu tolerance "
                         "SURVEY 8c's); f16x3g: f16x3 whose backward gate gradients travel as ONE fp16 plane (include/windgnn.h, "
                         "WGNN_MATH_F16X3G; reported as the labelled secondary `f16x3g_mixed`); f32: exact fp32; f16: one-pass fp16")
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"],
                    help="c3: 34 stations, the headline config; c5: 4096-station k-NN CSR stress config")
    ap.add_argument("--io", default="fp32", choices=["fp32", "fp16", "bf16"],
                    help="element type of X, Y and the labels on the wire (16-bit: BASELINE configs[2]; needs --math f16x3|f16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 PMC passes (N = 1 only)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the exact-fp32 secondary measurement")
    ap.add_argument("--no-c5", action="store_true", help="skip the 4096-station secondary (c5_csr_b128: ~40 GB of GPU memory, ~20 s)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--c5-child", default=None, choices=["f16x3", "f16x3g"], help=argparse.SUPPRESS)
    ap.add_argument("--direct-rccl", action="store_true",
                    help="N > 1: the gradient all-reduce on the compute stream through the step's own RCCL communicator "
                         "(distributed.DirectRccl; opt-in, also WGNN_RCCL_DIRECT=1) instead of torch.distributed's stream")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group and run the collective step even with one rank (rehearses the "
                         "N > 1 code path on a one-GPU box; launch with torch.distributed.run --nproc-per-node 1)")
    args = ap.parse_args()

    # RCCL on this pool needs dmabuf IPC (HSA_ENABLE_IPC_MODE_LEGACY=0): exported here, before anything initialises the
    # GPU, so the line does not depend on the launcher's environment (windgnn_amd/distributed.py)
    from windgnn_amd.distributed import ensure_rccl_env
    ensure_rccl_env()

    if args.c5_child:                                   # child of c5_in_child(): one dict on stdout, nothing else
        torch.cuda.set_device(0)
        print(json.dumps(secondary_c5(torch.device("cuda", 0), args.c5_child)), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d != WORLD_SIZE %d" % (args.gpus, world))
    ndev = torch.cuda.device_count()                    # does not initialise the GPU
    if local_rank >= ndev:
        raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPU(s) visible: one process per GPU, launch with "
                         "--nproc-per-node <= %d" % (local_rank, ndev, ndev))

    traffic, traffic_src = {}, None
    if world == 1 and not args.traffic_child:
        if not args.no_traffic:
            traffic, traffic_src = measure_traffic_live(args)
        if not traffic:
            why = traffic_src
            traffic, traffic_src = (({}, None) if args.workload != "c3" or args.math != "f16x3" or args.io != "fp32"
                                    else committed_traffic())
            if traffic_src:
                traffic_src = "committed file %s (%s)" % (traffic_src, why or "--no-traffic")

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:                                  # --force-dist without a launcher: a one-rank group on this GPU
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
                os.environ.setdefault(k, v)
        dist.init_process_group("nccl", device_id=dev)

    from windgnn_amd import GCN_GRU, _lib
    from windgnn_amd.trainer import TrainStep

    torch.manual_seed(0)                                # identical parameters on every rank
    S, H = (34, 102) if args.workload == "c3" else (4096, 12288)
    B = args.batch or (4096 if args.workload == "c3" else 128)
    model = GCN_GRU(F, F, F, S * F, H, math=args.math).to(dev)
    trainer = TrainStep(model, process_group=dist.group.WORLD if use_dist else None,
                        direct_rccl=True if args.direct_rccl else None)
    # an all-reduce of ones through the path the step's bucket takes: the ranks that really exchange data
    ranks_seen = trainer.exchange.ranks_seen() if use_dist else None
    A = (adjacency_34() if args.workload == "c3" else adjacency_knn(S)).to(dev)
    X, L = make_inputs(B, rank, dev, S, H, args.io)
    esz = 4.0 if args.io == "fp32" else 2.0

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- measurement passes that are not the timed region come FIRST (every rank runs them, rank 0 keeps the numbers):
    # the per-kernel pass (hipEvents inside the library) and the forward-only timing.  A 20 + 20 step run is 33 ms of GPU
    # time; entered cold it times the clock ramp (round 1: 4.59 M windows/s in the driver's 20-step run vs 4.85 M over 200
    # steps of the same code).  The timed region below is still exactly `warmup` untimed + `steps` timed steps.
    recs, fwd_s = None, None
    if not args.traffic_child:
        from windgnn_amd.functional import gcn_gru_forward_raw
        for _ in range(max(args.warmup, 20) if args.workload == "c3" else args.warmup):   # untimed: first launches, clock ramp
            trainer.step(A, X, L, world * B)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(args.steps):
            trainer.step(A, X, L, world * B)        # the kernels of the timed step, optimiser tail included
        torch.cuda.synchronize()
        recs = _lib.profile_read()
        _lib.profile_enable(False)
        check_stashless_forward(A, X, L, trainer, model.math)   # what is timed further down is the training forward's Y, bit for bit
        for _ in range(20):     # untimed: the stash-less forward runs kernels the training step does not (first launch, code load)
            gcn_gru_forward_raw(A, X, trainer.p_views, model.math, want_stash=False, prepared=trainer._prepared)
        # (the forward itself is timed AFTER the timed steps, on a GPU they have just kept busy: the check above ends in a host
        # synchronisation, the GPU idles and drops its clocks, and a short loop right behind it read 225-269 us for a forward that
        # takes 207-212 in a 200-step run: profiles/r5_bench_short_run_idle_gap.txt, r5_b_clock_ramp.txt)

    n_global = world * B                                # fixed global batch: the exchange needs no count collective
    import gc
    gc.collect()                                        # (tens of ms of host time: BEFORE the spin-up, or the GPU idles again)
    gc.disable()                                        # no collector pause inside a 15 ms timed region (as timeit does)
    if not args.traffic_child:
        # Untimed spin-up on every rank, before the contract's own W warm-up steps: the checks above synchronise with the host,
        # the GPU idles for a few ms and drops its clocks, and it takes ~25 training steps (20 ms) of load to get them back
        # (per-step times after a 5 ms idle gap: 773 794 822 848 860 852 830 808 ... 753 us against 745 steady:
        # profiles/r5_b_clock_ramp.txt).  With --steps 20 --warmup 5 the timed region would otherwise sit inside that ramp.
        nspin = SPINUP_STEPS if args.workload == "c3" else 2              # (a configs[4] step is 140 ms: two are a ramp's worth)
        spin_ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        spin_ev[0].record()
        for i in range(nspin):
            trainer.step(A, X, L, n_global)
            if i + 1 == nspin // 2:
                spin_ev[1].record()
        spin_ev[2].record()
    for _ in range(args.warmup):
        trainer.step(A, X, L, n_global)
    barrier()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    g0.record()
    for _ in range(args.steps):
        loss, _ = trainer.step(A, X, L, n_global)
    g1.record()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if args.traffic_child:
        return
    gpu_dt = g0.elapsed_time(g1) * 1e-3                # the same K steps on the GPU's clock: a host stall shows as dt >> gpu_dt
    h = max(nspin // 2, 1)
    spin_ms = [round(spin_ev[0].elapsed_time(spin_ev[1]) / h, 4), round(spin_ev[1].elapsed_time(spin_ev[2]) / max(nspin - h, 1), 4)] if nspin >= 2 else None
    if recs is not None:                                # forward-only timing, every rank (no collective inside), GPU still hot
        nfwd = max(args.steps, 50)
        for _ in range(10):
            gcn_gru_forward_raw(A, X, trainer.p_views, model.math, want_stash=False, prepared=trainer._prepared)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nfwd):
            gcn_gru_forward_raw(A, X, trainer.p_views, model.math, want_stash=False, prepared=trainer._prepared)
        e1.record()
        torch.cuda.synchronize()
        fwd_s = e0.elapsed_time(e1) * 1e-3 / nfwd
    trainer.check()                                     # fp16-plane modes: nothing left fp16's range
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- per-kernel pass (same workload, hipEvents inside the library) for the roofline object
    roofline, forward, kernels, mfma, path, secondary, extra = None, None, None, None, None, None, {}
    if rank == 0:
        recs.sort(key=lambda r: -r["ms"])
        kernels = []
        for r in recs:
            tr = traffic.get(r["name"])
            kernels.append({"name": r["name"], "launches_per_step": r["launches"] / args.steps,
                            "avg_us": round(1e3 * r["ms"] / max(r["launches"], 1), 2),
                            "ms_per_step": round(r["ms"] / args.steps, 4),
                            "algorithmic_bytes_per_launch": round(r["bytes"] / max(r["launches"], 1)),
                            "traffic_bytes_per_launch": round(tr["bytes_per_launch"]) if tr else None})
        d = recs[0]
        avg_s = d["ms"] / d["launches"] * 1e-3
        # which roofline binds this kernel: time its ALGORITHMIC bytes need at the HBM peak vs the time its
        # algorithmic flops need at the MFMA peak (f16x3 issues 3 MFMA passes per product => peak / 3)
        eff_peak = MFMA_PEAK_TFLOPS[args.math] / (3.0 if args.math in ("f16x3", "f16x3g") else 1.0)
        t_hbm = d["bytes"] / (HBM_PEAK_GBS * 1e9)
        t_mfma = d["flops"] / (eff_peak * 1e12)
        bound = "hbm" if (d["name"].startswith(BOUND_HBM_PREFIXES) or t_hbm >= t_mfma) else "mfma"
        tr = traffic.get(d["name"])
        trb = round(tr["bytes_per_launch"]) if tr else None
        if bound == "mfma":
            ach = d["flops"] / d["launches"] / avg_s / 1e12
            peak = MFMA_PEAK_TFLOPS[args.math]
            roofline = {"kernel": d["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": trb,
                        "avg_launch_us": round(avg_s * 1e6, 2)}
        else:
            ach = d["bytes"] / d["launches"] / avg_s / 1e9
            roofline = {"kernel": d["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": trb,
                        "avg_launch_us": round(avg_s * 1e6, 2)}
        roofline["algorithmic_bytes_per_launch"] = round(d["bytes"] / d["launches"])
        if d["name"].startswith("gcnx_bwd"):
            roofline["note"] = ("bound by its chain of dependent per-tile products, not by HBM or issue (DESIGN.md 5); in "
                                "f16x3g / f16 its dg operand is one fp16 plane (349 MB per launch instead of 436 MB)")
        roofline["traffic_source"] = traffic_src
        if trb:
            roofline["traffic_vs_algorithmic"] = round(trb / (d["bytes"] / d["launches"]), 3)
        # MFMA side of the picture (north-star: "HBM GB/s and MFMA utilisation against the roofline"): the GEMM with
        # the most time; f16x3 issues 3 fp16 MFMA passes per algorithmic FLOP
        gemms = [r for r in recs if r["name"].startswith(("pgemm_", "gemm_f32")) and r["flops"] > 0]
        if gemms:
            gk = gemms[0]
            passes = 1.0
            if gk["name"].startswith("pgemm_") and args.math in ("f16x3", "f16x3g"):
                passes = 2.0 if gk["name"].endswith(",x2>") else 3.0
            peak_k = MFMA_PEAK_TFLOPS["f32" if gk["name"].startswith("gemm_f32") else args.math]
            tf = gk["flops"] / (gk["ms"] * 1e-3) / 1e12
            mfma = {"kernel": gk["name"], "algorithmic_TFLOPs": round(tf, 1), "issued_TFLOPs": round(tf * passes, 1),
                    "peak": peak_k, "frac_issued": round(tf * passes / peak_k, 4)}
        # ---- the path as a whole: 2X + 2Y + L per window, fwd + bwd with recompute
        alg_step = float(B) * T * (2 * S * F + 3 * H) * esz
        step_s = dt / args.steps
        path = {"algorithmic_bytes_per_step": round(alg_step),
                "fwd_bwd_GBs": round(alg_step / step_s / 1e9, 1),
                "fwd_bwd_frac": round(alg_step / step_s / 1e9 / HBM_PEAK_GBS, 4)}
        if traffic:
            tot = sum(k["traffic_bytes_per_launch"] * k["launches_per_step"] for k in kernels
                      if k["traffic_bytes_per_launch"])
            path["traffic_bytes_per_step"] = round(tot)
            path["traffic_vs_algorithmic"] = round(tot / alg_step, 2)
            path["traffic_GBs"] = round(tot / step_s / 1e9, 1)
        # forward-only timing: north-star "fused forward vs HBM roofline" (52 224 algorithmic B/window)
        fwd_bytes = B * T * (S * F + H) * esz
        forward = {"us": round(fwd_s * 1e6, 1), "algorithmic_GBs": round(fwd_bytes / fwd_s / 1e9, 1),
                   "hbm_frac": round(fwd_bytes / fwd_s / 1e9 / HBM_PEAK_GBS, 4),
                   "windows_per_s": round(B / fwd_s, 1),
                   "checked": "Y of the timed stash-less call == the training forward's Y, bit for bit, in this run"}
        # ---- secondaries on the same box, each a labelled dtype of its own (never folded into `value`):
        #   f16x3g_mixed         the headline workload in WGNN_MATH_F16X3G (mixed precision in the backward; labelled, with its
        #                        measured gradient error next to it)
        #   exact_f32            the headline workload in WGNN_MATH_F32 (bitwise fp32 fmaf chains, fp32-input MFMA)
        #   c1_f32_b256          BASELINE configs[1]: B = 256, exact fp32 (the parity config)
        #   c2_f16_bf16io_b4096  BASELINE configs[2] literally: one-pass fp16 MFMA, bf16 X / Y / labels, B = 4096
        if world == 1 and args.math == "f16x3" and args.workload == "c3" and not args.no_secondary and args.io == "fp32" \
                and args.batch is None:
            extra["f16x3g_mixed"] = secondary_config(
                "f16x3g", "fp32", B, A, dev, max(5, min(args.steps, 30)),
                "same workload and step, WGNN_MATH_F16X3G: mixed precision -- the forward is f16x3 bit for bit, the backward's "
                "gate gradients travel as ONE fp16 plane and its GEMMs run two / one MFMA passes; NOT the parity-precision "
                "number")
            extra["f16x3g_mixed"]["worst_gradient_error_vs_fp64_oracle"] = F16X3G_GRAD_ERR
            secondary = secondary_config("f32", "fp32", B, A, dev, max(5, min(args.steps, 30)),
                                         "same workload and step, WGNN_MATH_F32 (fp32-input MFMA, bitwise fp32 fmaf chains)")
            extra["c1_f32_b256"] = secondary_config("f32", "fp32", 256, A, dev, max(10, min(2 * args.steps, 60)),
                                                    "BASELINE configs[1]: S=34, T=24, B=256, exact fp32 math and I/O")
            extra["c2_f16_bf16io_b4096"] = secondary_config(
                "f16", "bf16", 4096, A, dev, max(5, min(args.steps, 30)),
                "BASELINE configs[2]: S=34, T=24, B=4096, one-pass fp16 MFMA (fp32 accumulate), bf16 X / Y / labels; "
                "own tolerance (Y 2e-2), never the fp32-parity number", forward=True)
            if not args.no_c5:
                # ~40 GB of GPU memory and 2.4 G parameters per mode: each runs in a CHILD process with a time limit, so that
                # an out-of-memory kill, a GPU fault or a hang there costs this one field and not the headline measured above
                # (a try / except only catches Python exceptions)
                extra["c5_csr_b128"] = c5_in_child("f16x3")
                if "error" not in extra["c5_csr_b128"]:
                    mixed = c5_in_child("f16x3g")
                    extra["c5_csr_b128"]["f16x3g_mixed"] = {k: v for k, v in mixed.items()
                                                            if k in ("dtype", "value", "ms_per_step", "steps", "roofline", "error")}

    if rank == 0:
        out = {
            "metric": "windows/sec (fwd+bwd) %s seq24 GCN-GRU" % ("34-node" if args.workload == "c3" else "4096-node"),
            "value": round(world * B * args.steps / dt, 1),
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "ms_per_step_gpu_clock": round(1e3 * gpu_dt / args.steps, 4),   # same K steps between two events on the stream (diagnostic)
            "ms_per_step_spinup_halves": spin_ms,                            # the untimed spin-up's two halves (diagnostic: the clock ramp)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": (DTYPE_LABEL[args.math] if not (args.workload == "c5" and args.math == "f16x3g")
                      else "f16x3g (= f16x3 on the wide-GRU / CSR path: every product three-pass split fp16)") +
                     ("" if args.io == "fp32" else " math, %s I/O" % args.io),
            "data": "synthetic",
            "config": {"workload": "S=%d stations%s, T=24, F=13, H=%d, B=%d windows/GPU; step = forward + MSE + "
                                   "backward + grad all-reduce (N>1) + Adam; %s I/O"
                                   % (S, "" if args.workload == "c3" else " (symmetric 8-NN graph, CSR)", H, B, args.io),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "math": args.math,
                       "collective": (("one rccl all-reduce per step: [loss | conv | GRU gradients], 0.67 MB, between wgnn_finish(6) and "
                                       "wgnn_finish(0, adam)" + (", enqueued on the compute stream through the step's own communicator"
                                                                 if trainer.exchange.direct is not None else
                                                                 ", through torch.distributed (its own stream)" +
                                                                 ("; the own communicator was asked for and declined: %s"
                                                                  % trainer.exchange.direct_declined
                                                                  if trainer.exchange.direct_declined else "")))
                                      if use_dist else "none (one rank)")},
            "rccl_ranks_seen": ranks_seen,
            "loss": round(float(loss), 6),
            "roofline": roofline,
            "path": path,
            "mfma": mfma,
            "forward": forward,
            "f16x3g_mixed": extra.get("f16x3g_mixed"),
            "exact_f32": secondary,
            "c1_f32_b256": extra.get("c1_f32_b256"),
            "c2_f16_bf16io_b4096": extra.get("c2_f16_bf16io_b4096"),
            "c5_csr_b128": extra.get("c5_csr_b128"),
            "measurement_order": "traffic children (N=1), untimed per-kernel pass + forward output check, %d untimed spin-up steps (clock ramp after the host-synchronising checks), then `warmup` untimed and `steps` timed steps, forward-only timing, secondaries (f16x3g_mixed, exact_f32, c1, c2), cpu baseline" % SPINUP_STEPS,
            "kernels": kernels,
            "kernels_note": "hipEvent-bracketed inside the library: every launch carries 1-2 us of event latency, so the sum "
                            "runs 3-4 % above ms_per_step; the rocprofv3 durations (profiles/*_kernel_stats.csv) sum to it",
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "c3":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        trainer.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
