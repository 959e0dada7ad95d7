"""Build libmgx.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

Cross-compiles without a GPU.  The .so stays inside the package directory so that it travels
with the source tree (it is git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmgx.so")
SOURCES = ["api.cpp", "rowwise_ops.hip", "rel_attn_fwd.hip", "rel_attn_bwd.hip", "linear.hip", "decode.hip", "gru_train.hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def have_hipcc() -> bool:
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "mgx.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every source in csrc/ for gfx950 and link libmgx.so.  Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
              "-I" + CSRC, "-Wno-unused-value", "-Wno-unused-result"]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = common + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
