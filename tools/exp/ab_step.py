"""Same-box A/B of library builds on the training step at the bench shape: per-kernel event times + the step time.
    python tools/exp/ab_step.py [--math f16x3] [--rounds 2] [--batch 4096] libA.so libB.so ...
Each build runs in its own process (WGNN_LIB), builds alternate, so clock / box drift shows as spread between rounds."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from bench import adjacency_34, make_inputs, S, T, F, H
from windgnn_amd import GCN_GRU, _lib
from windgnn_amd.trainer import TrainStep
dev = torch.device("cuda:0")
math, B = sys.argv[1], int(sys.argv[2])
m = GCN_GRU(F, F, F, S * F, H, math=math).to(dev)
tr = TrainStep(m)
A = adjacency_34().to(dev)
X, L = make_inputs(B, 0, dev)
for _ in range(30): tr.step(A, X, L)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): tr.step(A, X, L)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 100
_lib.profile_enable(True)
for _ in range(20): tr.step(A, X, L)
torch.cuda.synchronize()
recs = _lib.profile_read()
print("step %%.1f us | " %% (dt * 1e6) + " ".join("%%s=%%.1f" %% (r["name"].replace("_kernel", "")[:22], 1e3 * r["ms"] / r["launches"]) for r in recs))
''' % ROOT
args = sys.argv[1:]
math, rounds, batch = "f16x3", 2, 4096
libs = []
while args:
    a = args.pop(0)
    if a == "--math": math = args.pop(0)
    elif a == "--rounds": rounds = int(args.pop(0))
    elif a == "--batch": batch = int(args.pop(0))
    else: libs.append(os.path.abspath(a))
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, WGNN_LIB=lib)
        out = subprocess.run([sys.executable, "-c", CHILD, math, str(batch)], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("step")]
        print("%-14s %s" % (os.path.basename(lib), line[0] if line else "FAILED: " + out.stderr[-400:]), flush=True)
