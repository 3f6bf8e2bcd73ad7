"""Host AddressSanitizer check of the C-ABI library (CPU box only -- GPU ASan is not available on the pool).

Builds every csrc/*.hip into build/asan/libwindgnn_hip_asan.so with `-fsanitize=address -fno-gpu-sanitize` (host code
instrumented, device code untouched) and drives, in a child interpreter with the ASan runtime preloaded, every entry
point's host-side logic that runs without a GPU: layout / size computation for many shapes (including BASELINE's
4096-station configuration), argument validation, error strings, the profile aid.  Any ASan report fails the check.

    python tools/asan_host_check.py          # exit code 0 = clean
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "windgnn_amd", "csrc")
OUT = os.path.join(ROOT, "build", "asan")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

DRIVER = r'''
import ctypes as C, sys
sys.path.insert(0, %(root)r)
from windgnn_amd import _lib as L
lib = C.CDLL(%(so)r)
for name, (res, args) in L.EXPORTS.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
assert lib.wgnn_version() >= 122
for st in range(-9, 2):
    assert lib.wgnn_strerror(st)
n = 0
for B in (1, 2, 15, 16, 17, 256, 4096):
    for (S, H) in ((1, 1), (3, 9), (7, 21), (34, 102), (48, 33), (64, 127), (20, 200), (65, 12), (4096, 12288)):
        for math in (0, 1, 2, 3, 7):
            for fmt, nnz in ((0, 0), (1, 9 * S), (1, 0), (3, 0)):
                for io in (0, 1, 2, 5):
                    d = L.Dims(B, 24, S, 13, H, math, fmt, nnz, io)
                    w, s = lib.wgnn_workspace_bytes(C.byref(d)), lib.wgnn_stash_bytes(C.byref(d))
                    assert (w == 0) == (s == 0), (B, S, H, math, fmt, io)
                    n += 1
                    p, g, ad = L.Params(), L.Grads(), L.Adam()
                    pb = lib.wgnn_prepared_bytes(C.byref(d))
                    assert pb == 0 or w > 0
                    ad.step = 1
                    for which, adam in ((6, None), (4, None), (2, None), (0, ad), (6, ad), (8, None), (0, None)):
                        assert lib.wgnn_finish(C.byref(d), C.byref(p), C.byref(g), which, C.byref(adam) if adam else None, None, 0, None) < 0
                    assert lib.wgnn_prepare_weights(C.byref(d), C.byref(p), None, 0, None) < 0
                    for fn in (lambda: lib.wgnn_fwd(C.byref(d), None, None, C.byref(p), None, None, None, 0, None),
                               lambda: lib.wgnn_fwd_loss(C.byref(d), None, None, C.byref(p), None, None, None, None, 0, None),
                               lambda: lib.wgnn_fwd_last(C.byref(d), None, None, C.byref(p), 0.0, 1.0, None, None, 0, None),
                               lambda: lib.wgnn_bwd(C.byref(d), None, None, C.byref(p), None, None, None, C.byref(g), None, 0, None),
                               lambda: lib.wgnn_bwd_part(C.byref(d), None, None, C.byref(p), None, None, None, C.byref(g), None, 0, None, 9),
                               lambda: lib.wgnn_bwd_mse_part(C.byref(d), None, None, C.byref(p), None, None, 1.0, None, None, C.byref(g), None, 0, None, 15)):
                        assert fn() < 0
for bad in (L.Dims(0, 1, 1, 13, 1, 0, 0, 0, 0), L.Dims(1, 1, 1, 12, 1, 0, 0, 0, 0), L.Dims(1 << 30, 4, 1, 13, 1, 0, 0, 0, 0)):
    assert lib.wgnn_workspace_bytes(C.byref(bad)) == 0
assert lib.wgnn_workspace_bytes(None) == 0 and lib.wgnn_stash_bytes(None) == 0
assert lib.wgnn_gcn_layer_workspace_bytes(10, 34, 13, 13) > 0 and lib.wgnn_gcn_layer_workspace_bytes(10, 65, 13, 13) == 0
assert lib.wgnn_gcn_layer_workspace_bytes(10, 34, 5, 64) > 0 and lib.wgnn_gcn_layer_workspace_bytes(10, 34, 65, 13) == 0
assert lib.wgnn_gcn_layer_csr_workspace_bytes(10, 4096, 13) > 0
assert lib.wgnn_gcn_layer_fwd(4, 34, 13, 13, None, None, None, None, None, None) == -1
assert lib.wgnn_gcn_layer_fwd(4, 34, 6, 9, None, None, None, None, None, None) == -1
assert lib.wgnn_gcn_layer_fwd(4, 34, 6, 65, None, None, None, None, None, None) == -5
assert lib.wgnn_gcn_layer_bwd(4, 34, 13, 13, None, None, None, None, None, None, None, None, None, 0, None) == -1
dg = L.Dims(4, 24, 34, 13, 102, 0, 0, 0, 0)
assert lib.wgnn_gru_fwd(C.byref(dg), None, C.byref(L.Params()), None, None, None, 0, None) == -1
assert lib.wgnn_gru_bwd(C.byref(dg), None, C.byref(L.Params()), None, None, None, C.byref(L.Grads()), None, None, 0, None) == -1
for k in range(-1, 7):
    lib.wgnn_get_option(k); lib.wgnn_set_option(k, 9)
assert lib.wgnn_set_option(0, 2) in (0, 1, 2) and lib.wgnn_get_option(0) == 2 and lib.wgnn_set_option(0, 1) == 2
assert lib.wgnn_gcn_layer_csr_fwd(4, 34, 13, 0, None, None, None, None, None, None) == -2
assert lib.wgnn_mse_loss_grad(None, None, 5, 1.0, None, None, None, 0, None) == -1
assert lib.wgnn_adam_step(None, None, None, None, 5, 1, 1e-3, 0.9, 0.999, 1e-8, None) == -1
assert lib.wgnn_make_windows(None, 100, 7, 13, 12, 11, None, None, 4, None, None, None) == -1
assert lib.wgnn_predict_last(None, 1, 1, 1, 0.0, 1.0, None, None) == -1
assert lib.wgnn_profile_enable(0) == 0
print("asan host check: %%d layouts, all entry points validated, no report" %% n)
'''


def main() -> int:
    os.makedirs(OUT, exist_ok=True)
    from windgnn_amd.build import SOURCES
    flags = ["-O1", "-g", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fsanitize=address", "-fno-gpu-sanitize",
             "-mllvm", "-amdgpu-mfma-vgpr-form"]
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OUT, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen([HIPCC, *flags, "-c", os.path.join(CSRC, src), "-o", obj], stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for src, p in zip(SOURCES, procs):
        out, _ = p.communicate()
        if p.returncode != 0:
            print("hipcc (asan) failed on %s:\n%s" % (src, out))
            return 2
    so = os.path.join(OUT, "libwindgnn_hip_asan.so")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize", *objs,
                    "-o", so], check=True)
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "--print-file-name=libclang_rt.asan-x86_64.so"],
                        stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.exists(rt):
        print("ASan runtime not found: %s" % rt)
        return 3
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",
               PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT, "so": so}], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    print(r.stdout[-3000:])
    if r.returncode != 0 or "AddressSanitizer" in r.stdout:
        return r.returncode or 1
    return 0


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.exit(main())
