"""Build libwindgnn_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m windgnn_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libwindgnn_hip.so")
SOURCES = ["api.hip", "finish.hip", "gcn.hip", "gemm.hip", "gru.hip", "train_ops.hip", "prof.hip", "gcnx.hip", "grux.hip", "pgemm.hip", "data_ops.hip", "general.hip", "gru_small.hip", "gcn32.hip", "gemm32.hip", "gcngi.hip", "pgemm_big.hip", "gcn_any.hip"]
HEADERS = ["common.h", "gcnx_dev.h", os.path.join("..", "..", "include", "windgnn.h")]
# -amdgpu-mfma-vgpr-form: MFMA accumulators live in VGPRs, so no v_accvgpr_read per value in the VALU-bound
# GCN/GRU kernels (gcnx_bwd -8 %).  Safe only because every first read of an MFMA result is a compiler-visible
# instruction (see split2 in gcnx.hip).
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]
FLAGS += os.environ.get("WGNN_HIPCC_FLAGS", "").split()   # experiments only; the default build takes none


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print("hipcc failed on %s:\n%s" % (src, out), file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
