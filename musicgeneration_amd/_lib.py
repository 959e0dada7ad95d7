"""ctypes binding of libmgx.so (include/mgx.h).  No fallback: if the library is missing or a call
fails, the product path raises -- there is no CPU/eager substitute (the oracle lives in oracle/ and
is never imported from here)."""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGX_LIB_PATH") or os.path.join(PKG, "libmgx.so")      # override: A/B two builds in one run

_vp, _i, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t

# the ABI the SIGNATURES table below was written for (MGX_ABI_VERSION of include/mgx.h).  A left-over
# libmgx.so of another ABI still exports the same names: calling it with this table would shift arguments.
EXPECTED_ABI = 18

# name -> argtypes ; every symbol declared in include/mgx.h (restype int unless noted)
SIGNATURES = {
    "mgx_abi_version": [],
    "mgx_device_count": [],
    "mgx_set_deterministic": [_vp, _sz],
    "mgx_deterministic": [],
    "mgx_set_deterministic_stream": [_vp, _vp, _sz],
    "mgx_stream_create_cu_mask": [_vp, _vp, _i],
    "mgx_stream_set_cus": [_vp, _i],
    "mgx_stream_cus": [_vp],
    "mgx_stream_destroy": [_vp],
    "mgx_linear_kernel_id": [_i, _i, _i, _i, _vp],
    "mgx_embed_pe_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _vp],
    "mgx_embed_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _vp],
    "mgx_pad_bitmap": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mgx_rel_attn_fwd_workspace": [_i],                  # returns size_t
    "mgx_rel_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp],
    "mgx_rel_attn_fwd_nomask": [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp],
    "mgx_rel_attn_weights": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp],
    "mgx_rel_attn_bwd_workspace": [_i, _i, _i],          # returns size_t
    "mgx_rel_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp],
    "mgx_rel_attn_bwd_parts": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp],
    "mgx_add_ln_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _u64, _vp],
    "mgx_add_ln_bwd_workspace": [_i, _i],                 # returns size_t
    "mgx_add_ln_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _f, _u64, _vp],
    "mgx_smooth_ce_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "mgx_smooth_ce_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _vp, _vp],
    "mgx_adam_step": [_vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _i, _f, _vp],
    "mgx_cast_bf16": [_vp, _vp, _sz, _vp],
    "mgx_linear_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mgx_decode_embed": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mgx_rel_attn_decode_workspace": [_i, _i, _i],       # returns size_t
    "mgx_rel_attn_decode": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp],
    "mgx_sample_topk_topp": [_vp, _i, _i, _f, _i, _f, _u64, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp],
    "mgx_sample_topk_topp_rows": [_vp, _i, _i, _f, _i, _f, _u64, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp],
    "mgx_gather_rows": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mgx_gru_gates": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "mgx_linear_dx": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mgx_gru_cell_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "mgx_gru_cell_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "mgx_gru_step_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "mgx_gru_step_x_fwd": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "mgx_gru_step_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mgx_dropout_bf16": [_vp, _vp, _sz, _f, _u64, _vp],
    "mgx_scatter_add_rows": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mgx_linear_ln_fwd": [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mgx_decode_embed_linear": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mgx_decode_embed_linear_frag": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mgx_skinny_fwd_frag": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mgx_linear_ln_fwd_frag": [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mgx_rel_attn_decode_splits": [_i, _i, _i],
    "mgx_linear_dw_grouped": [_vp, _i, _i, _vp, _sz, _vp],
    "mgx_linear_dw_grouped_workspace": [_vp, _i, _i],     # returns size_t
    "mgx_linear_dw": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
}

_lib = None


class MgxError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libmgx.so (building it first if hipcc is available and the .so is absent/stale)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must load ITS libamdhip64 first: libmgx.so then binds to the same HIP runtime instance
    # (loading libmgx first would pull in /opt/rocm's copy and leave two runtimes in one process).
    import torch  # noqa: F401
    own = os.path.abspath(LIB_PATH) == os.path.join(PKG, "libmgx.so")
    if own:
        from . import _build
        force = os.environ.get("MGX_REBUILD") == "1"
        if force or not os.path.exists(LIB_PATH) or (_build.have_hipcc() and _build._stale()):
            # absent or older than a source / the header: rebuild when a compiler is here (the GPU box has one too)
            if _build.have_hipcc():
                _build.build(force=force)
    if not os.path.exists(LIB_PATH):
        raise MgxError(f"{LIB_PATH} is missing: run `python -m musicgeneration_amd._build` (needs hipcc)")
    lib = C.CDLL(LIB_PATH)
    lib.mgx_last_error.restype = C.c_char_p
    lib.mgx_last_error.argtypes = []
    lib.mgx_abi_version.restype = C.c_int
    lib.mgx_abi_version.argtypes = []
    abi = lib.mgx_abi_version()
    if abi != EXPECTED_ABI:
        raise MgxError(f"{LIB_PATH} has ABI version {abi}, this package binds ABI {EXPECTED_ABI} (include/mgx.h): "
                       "stale build -- rebuild with `python -m musicgeneration_amd._build --force`")
    for name, argt in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch: fail loudly
        fn.restype = C.c_size_t if name.endswith("_workspace") else C.c_int
        fn.argtypes = argt
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().mgx_last_error().decode("utf-8", "replace")
        raise MgxError(f"{what} failed (status {rc}): {msg}")


def ptr(t) -> int:
    """device pointer of a torch tensor (or None -> NULL)"""
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream_ptr() -> int:
    """raw hipStream_t of torch's current stream on the current device.  Through torch's C entry point: the Python-level
    ``torch.cuda.current_stream().cuda_stream`` costs ~8 us per call (device-index bookkeeping, a Stream object), i.e. 0.4 ms of
    a training step's ~150 launches -- nothing at the bench shape, a fifth of the host's time per step where the step is a few
    milliseconds (d_model 256, batch 2-6: round 6, profiles/r06_host_overhead.txt)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        raw, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        _raw_stream = (lambda: raw(dev())) if (raw is not None and dev is not None) else (lambda: torch.cuda.current_stream().cuda_stream)
    return _raw_stream()
