"""Build libmgx.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

Cross-compiles without a GPU.  The .so stays inside the package directory so that it travels
with the source tree (it is git-ignored, not gpurun-ignored).

`python -m musicgeneration_amd._build [--force]` builds the product library.
`python -m musicgeneration_amd._build --variant NAME -DMACRO[=v] ...` builds `libmgx_NAME.so` beside it with extra
macros: the timing-only "peel" / in-kernel "stamp" builds the profile notes quote are made this way, from the
tracked sources, and loaded with `MGX_LIB_PATH=musicgeneration_amd/libmgx_NAME.so` (tools/peel_*.sh, tools/ab.sh).
`--experiments` additionally compiles the alternative kernels kept under tools/experiments/ (two forward-attention
structures and the 64-keys-per-wave dK/dV kernel, all measured slower) and defines MGX_EXPERIMENTS=1, which is also what enables the environment knobs
(MGX_ATTN_FWD64, MGX_FWD_LDS, MGX_DKV_LDS, MGX_ATTN_BGROUP): the product library reads none of them."""
from __future__ import annotations

import fcntl
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmgx.so")
SOURCES = ["api.cpp", "rowwise_ops.hip", "rel_attn_fwd.hip", "rel_attn_bwd.hip", "rel_attn_dkv64.hip", "linear.hip", "decode.hip", "gru_train.hip"]
EXPERIMENT_DIR = os.path.join(ROOT, "tools", "experiments")
EXPERIMENT_SOURCES = ["rel_attn_fwd2.hip", "rel_attn_fwd3.hip", "rel_attn_fwd64.hip"]      # --experiments builds only
# per-file flags.  The 64-rows-per-wave attention kernels run one wave per SIMD with the whole 512-entry register file:
# MFMA results that VALU code reads (scores) must stay in arch VGPRs (with more than 256 registers available hipcc otherwise
# gives every MFMA an AGPR destination and copies each result out), and the SLP vectoriser must not pair the two blocks'
# row sums into v_pk_add_f32 (slower than two v_add_f32 beside MFMAs).
_W64 = ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"]
EXTRA_FLAGS = {"rel_attn_fwd2.hip": _W64,
               }


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


def _stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "mgx.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _generate(experiments: bool = False) -> None:
    """the hand-scheduled main loops are generated code: csrc/gen_dkv_asm.py -> csrc/rel_attn_dkv64_loop.inc (tracked, so the
    schedule can be read and diffed) and ..._loop_stamp.inc (diagnostic builds; not tracked); csrc/gen_gemm_asm.py ->
    csrc/linear_dw_ring4_loop.inc.  Regenerated on every build."""
    gens = [os.path.join(CSRC, "gen_dkv_asm.py"), os.path.join(CSRC, "gen_gemm_asm.py")] + ([os.path.join(EXPERIMENT_DIR, "gen_fwd_asm.py")] if experiments else [])
    for gen in gens:
        r = subprocess.run([sys.executable, gen], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(gen + " failed:\n" + r.stdout)


def _compile_and_link(lib: str, objdir: str, defines, verbose: bool, experiments: bool = False) -> None:
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)
    _generate(experiments)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
              "-I" + CSRC, "-Wno-unused-value", "-Wno-unused-result"] + list(defines)
    objs, procs = [], []
    pid = os.getpid()
    if experiments:
        common.append("-DMGX_EXPERIMENTS=1")
    todo = [(s_, os.path.join(CSRC, s_)) for s_ in SOURCES]
    if experiments:
        todo += [(s_, os.path.join(EXPERIMENT_DIR, s_)) for s_ in EXPERIMENT_SOURCES]
    for src, sp in todo:
        if not os.path.exists(sp):
            continue
        # pid-unique object names: two processes that get past the lock in turn never share a half-written object
        obj = os.path.join(objdir, f"{os.path.splitext(src)[0]}.{pid}.o")
        cmd = common + EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    try:
        for src, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{out}")
            if verbose and out.strip():
                print(out)
        tmp = f"{lib}.{pid}.tmp"
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
        os.replace(tmp, lib)                             # atomic: a concurrent CDLL sees the old or the new file, never a part
    finally:
        for o in objs:
            if os.path.exists(o):
                os.remove(o)


def build(force: bool = False, verbose: bool = False, variant: str | None = None, defines=(), experiments: bool = False) -> str:
    """Compile every source in csrc/ for gfx950 and link libmgx.so (or libmgx_<variant>.so with extra -D macros).
    Returns the library path.  Safe to call from several processes at once (torchrun ranks, bench.py's self-launch):
    the stale check and the build run under an exclusive file lock, and a rank that waited finds the library fresh."""
    if experiments and not variant:
        raise ValueError("--experiments needs --variant NAME: the product library is built without the experiment kernels")
    lib = LIB if not variant else os.path.join(PKG, f"libmgx_{variant}.so")
    if not force and not variant and not _stale(lib):
        return lib
    objdir = os.path.join(PKG, "build", variant or "product")
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    with open(os.path.join(PKG, "build", ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if force or variant or _stale(lib):          # re-check: another process may have built it while we waited
                _compile_and_link(lib, objdir, defines, verbose, experiments)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return lib


if __name__ == "__main__":
    argv = sys.argv[1:]
    var = argv[argv.index("--variant") + 1] if "--variant" in argv else None
    print(build(force="--force" in argv, verbose="-v" in argv, variant=var, defines=[a for a in argv if a.startswith("-D")],
                experiments="--experiments" in argv))
