"""Host-side wrappers over the C ABI (include/mgx.h): thin launch helpers + torch.autograd glue.

PyTorch is used for device memory, streams and autograd bookkeeping only; every op below runs a
hand-written HIP kernel from libmgx.so on the current HIP stream.  Nothing here falls back to
eager PyTorch: a missing library or a non-CUDA tensor raises.
"""
from __future__ import annotations

from typing import Tuple

import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

BF16 = torch.bfloat16


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.MgxError("mgx ops need CUDA/HIP tensors (no CPU fallback); got a CPU tensor")
        if t is not None and not t.is_contiguous():
            raise _lib.MgxError("mgx ops need contiguous tensors")


# --------------------------------------------------------------------------------------------------
# deterministic-reduction mode (include/mgx.h: mgx_set_deterministic)
# --------------------------------------------------------------------------------------------------
_det_scratch = None          # the registered buffer must outlive the mode: kept here
_det_stream_scratch = {}     # stream pointer -> buffer of that stream (mgx_set_deterministic_stream)
_DET_BYTES = 32 << 20        # include/mgx.h: 16 MiB of sums + 16 MiB of poison words, aligned to 32 MiB


def _det_buffer(dev):
    """-> (base tensor to keep alive, 32 MiB-aligned data pointer inside it)"""
    base = torch.empty(2 * _DET_BYTES, dtype=torch.uint8, device=dev)
    return base, base.data_ptr() + (-base.data_ptr()) % _DET_BYTES


def set_deterministic(on: bool = True, device=None) -> None:
    """Switch the library's cross-workgroup sums (dE, vocabulary dW / db, block bias gradients, embedding gradient, loss
    statistics) to order-independent fixed-point integer atomics: repeated runs -- and a data-parallel run against the
    single-process run of the same global batch -- then agree bit for bit on everything the kernels compute.  Costs a few
    percent; ``MGX_DETERMINISTIC=1`` in the environment switches it on when the first model moves to the GPU.  Process-wide.
    The scratch registered here serves ONE stream at a time; the CU-masked side stream of ``configure_streams`` gets a buffer
    of its own (registered here when it exists already, by ``configure_streams`` when it is created later)."""
    global _det_scratch
    lib = _lib.load()
    if not on:
        check(lib.mgx_set_deterministic(None, 0), "mgx_set_deterministic")
        _det_scratch = None
        _det_stream_scratch.clear()
        return
    dev = device if device is not None else torch.device("cuda")
    buf, p = _det_buffer(dev)
    check(lib.mgx_set_deterministic(p, _DET_BYTES), "mgx_set_deterministic")
    _det_scratch = buf
    for plan in _STREAM_PLANS.values():
        _det_register_stream(plan.side)


def _det_register_stream(ms) -> None:
    """deterministic mode on: the masked stream ``ms`` issues calls that use the fixed-point scratch -> give it one of its own"""
    if ms is None or not deterministic() or ms.ptr in _det_stream_scratch:
        return
    buf, p = _det_buffer(ms.device)
    check(_lib.load().mgx_set_deterministic_stream(ms.ptr, p, _DET_BYTES), "mgx_set_deterministic_stream")
    _det_stream_scratch[ms.ptr] = buf


def deterministic() -> bool:
    return bool(_lib.load().mgx_deterministic())


# --------------------------------------------------------------------------------------------------
# CU-masked streams (include/mgx.h: mgx_stream_create_cu_mask) and the two-stream plan of the backward
# --------------------------------------------------------------------------------------------------
class MaskedStream:
    """A HIP stream whose kernels may only run on ``cus`` CUs: logical CUs ``first .. first + cus - 1`` of the driver's
    numbering, which deals consecutive CUs round the XCDs first -- so a run of n (a multiple of 8) is n/8 CUs of every XCD, and
    two disjoint runs are disjoint sets of CUs.  ``.stream`` is the ``torch.cuda.ExternalStream`` over it (events, allocator,
    ``with torch.cuda.stream(...)``), ``.ptr`` the raw ``hipStream_t`` the library's CU-count registry knows."""

    def __init__(self, cus: int, first: int = 0, device=None):
        lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        total = torch.cuda.get_device_properties(self.device).multi_processor_count
        if cus <= 0 or first < 0 or first + cus > total:
            raise ValueError(f"MaskedStream: CUs {first}..{first + cus - 1} do not fit the device's {total}")
        words = (total + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        for i in range(first, first + cus):
            mask[i // 32] |= 1 << (i % 32)
        out = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.mgx_stream_create_cu_mask(ctypes.byref(out), ctypes.cast(mask, ctypes.c_void_p), words), "mgx_stream_create_cu_mask")
        self.ptr, self.cus, self.first = out.value, cus, first
        self.stream = torch.cuda.ExternalStream(self.ptr, device=self.device)

    def close(self, destroy: bool = False):
        """forget the stream.  The HIP stream itself is only destroyed on request: the caching allocator keeps the stream of every
        block that was allocated on it or marked with ``record_stream`` and records an event on it when such a block is freed --
        long after this call, possibly at interpreter shutdown -- so destroying a stream that tensors have touched is a
        use-after-free.  A forgotten stream costs one idle hardware queue."""
        if self.ptr is not None:
            self.stream.synchronize()
            if destroy:
                _det_stream_scratch.pop(self.ptr, None)
                check(_lib.load().mgx_stream_destroy(self.ptr), "mgx_stream_destroy")
            self.ptr = None


_MASKED = {}                 # (device index, first, cus) -> MaskedStream: created once, reused by every later plan


def masked_stream(cus: int, first: int = 0, device=None) -> "MaskedStream":
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, first, cus)
    ms = _MASKED.get(key)
    if ms is None or ms.ptr is None:
        ms = _MASKED[key] = MaskedStream(cus, first, dev)
    return ms


class StreamPlan:
    """``side``: the masked stream of the backward's off-critical-path kernels (dE from the stored dS tiles, the block's weight
    gradients -- they feed only the optimiser); ``main``: the masked stream of everything else (the complement of ``side``
    minus the CUs left to RCCL), or None when the caller's own stream stays unrestricted."""
    __slots__ = ("side", "main", "reserved", "last", "work")

    def __init__(self, side, main, reserved, work=3):
        self.side, self.main, self.reserved, self.last, self.work = side, main, reserved, None, work


_STREAM_PLANS = {}           # device index -> StreamPlan


def configure_streams(side_cus: int = 0, reserve_cus: int = 0, partition: bool = True, device=None, side_work: int = 3):
    """Run the HBM-bound, off-critical-path half of an encoder block's backward (``rel_attn_de_tiles_kernel`` and the grouped
    weight-gradient launch: ~0.8 of a block's ~3 ms at the bench shape, none of it needed before the optimiser step / the
    bucket's all-reduce) on a side stream restricted to ``side_cus`` CUs, BESIDE the MFMA-bound kernels of the critical path
    instead of between them.  ``partition``: also create the main stream, restricted to the other CUs -- run the training step
    inside ``with torch.cuda.stream(ops.main_stream())``; without it the caller's stream keeps the whole chip and the
    dispatcher shares the side stream's CUs between the two.  ``reserve_cus``: CUs (the highest-numbered ones) that neither
    stream may use -- left to RCCL's kernels under data parallelism.  ``side_cus = 0`` with
    ``reserve_cus > 0``: only the main stream, masked.  Both zero: back to one unrestricted stream.  ``side_work``: what goes to
    the side stream -- bit 0 the dE kernel, bit 1 the weight gradients.  Returns the plan (or None)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if _STREAM_PLANS.pop(idx, None) is not None:
        torch.cuda.synchronize(dev)                        # (the plan's streams stay alive in _MASKED: see MaskedStream.close)
    if side_cus <= 0 and reserve_cus <= 0:
        return None
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    if side_cus % 8 or reserve_cus % 8 or side_cus + reserve_cus >= total:
        raise ValueError(f"configure_streams: side_cus and reserve_cus must be multiples of 8 (whole-XCD-balanced masks) summing to "
                         f"less than {total}")
    side = masked_stream(side_cus, 0, dev) if side_cus > 0 else None
    main = masked_stream(total - side_cus - reserve_cus, side_cus, dev) if (partition or side is None) else None
    if side is not None and side_work not in (1, 2, 3):
        raise ValueError("configure_streams: side_work is 1 (dE), 2 (weight gradients) or 3 (both)")
    plan = _STREAM_PLANS[idx] = StreamPlan(side, main, reserve_cus, side_work)
    _det_register_stream(side)
    return plan


def stream_plan(device=None):
    idx = torch.cuda.current_device() if device is None or torch.device(device).index is None else torch.device(device).index
    return _STREAM_PLANS.get(idx)


def main_stream(device=None):
    """the stream a training step should run on: the plan's masked main stream, else the current stream"""
    plan = stream_plan(device)
    return plan.main.stream if plan is not None and plan.main is not None else torch.cuda.current_stream(device)


def join_side_stream(device=None) -> None:
    """the current stream waits for everything issued on the side stream (called before the optimiser step, and by anything else
    that reads the gradients the side stream accumulates)"""
    plan = stream_plan(device)
    if plan is not None and plan.side is not None and plan.last is not None:
        torch.cuda.current_stream(device).wait_event(plan.last)
        plan.last = None


# --------------------------------------------------------------------------------------------------
# raw launchers (no autograd)
# --------------------------------------------------------------------------------------------------
def pad_bitmap(tok: torch.Tensor, pad: int, flag: torch.Tensor | None = None) -> torch.Tensor:
    """tok int32 [B,L] -> uint32 bitmap [B, L/32] (stored as int32).  ``flag`` (int32[1] on the device, optional) gets bit 0
    OR-ed in when a row starts with a pad token and holds a real token later (leading padding: fully masked queries; sticky;
    read it at a synchronisation point)."""
    _need_cuda(tok, flag)
    B, L = tok.shape
    bits = torch.empty(B, L // 32, dtype=torch.int32, device=tok.device)
    check(_lib.load().mgx_pad_bitmap(ptr(tok), ptr(bits), ptr(flag), B, L, pad, stream_ptr()), "mgx_pad_bitmap")
    return bits


def embed_pe_fwd(tok, table, pe, p_drop=0.0, seed=0):
    _need_cuda(tok, table, pe)
    B, L = tok.shape
    V, d = table.shape
    out = torch.empty(B, L, d, dtype=BF16, device=tok.device)
    check(_lib.load().mgx_embed_pe_fwd(ptr(tok), ptr(table), ptr(pe), ptr(out), B, L, d, V, float(p_drop),
                                       int(seed), stream_ptr()), "mgx_embed_pe_fwd")
    return out


def embed_bwd(tok, dout, dtable, p_drop=0.0, seed=0):
    _need_cuda(tok, dout, dtable)
    B, L = tok.shape
    V, d = dtable.shape
    check(_lib.load().mgx_embed_bwd(ptr(tok), ptr(dout), ptr(dtable), B, L, d, V, float(p_drop), int(seed),
                                    stream_ptr()), "mgx_embed_bwd")


def rel_attn_fwd(qkv, E, padbits, M=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """qkv bf16 [B,L,3d], E bf16 [M,64], padbits int32 [B,L/32] or None -> (ctx bf16 [B,L,d], lse f32 [B,h,L])"""
    _need_cuda(qkv, E, padbits)
    B, L, d3 = qkv.shape
    d = d3 // 3
    M = E.shape[0] if M is None else M
    ctx = torch.empty(B, L, d, dtype=BF16, device=qkv.device)
    lse = torch.empty(B, d // 64, L, dtype=torch.float32, device=qkv.device)
    lib = _lib.load()
    ws = torch.empty(lib.mgx_rel_attn_fwd_workspace(L), dtype=torch.uint8, device=qkv.device)
    check(lib.mgx_rel_attn_fwd(ptr(qkv), ptr(E), ptr(padbits), ptr(ctx), ptr(lse), ptr(ws), ws.numel(), B, L, d, M,
                               stream_ptr()), "mgx_rel_attn_fwd")
    return ctx, lse


def rel_attn_fwd_nomask(qkv, E, Lk=None) -> torch.Tensor:
    """the reference's sampling call (mask=None): bidirectional attention over keys 0..Lk-1, relative term for j <= i only
    (mgx.h).  qkv bf16 [B,L,3d] -> ctx bf16 [B,L,d]; rows >= Lk are don't-cares.  Inference only."""
    _need_cuda(qkv, E)
    B, L, d3 = qkv.shape
    d = d3 // 3
    ctx = torch.empty(B, L, d, dtype=BF16, device=qkv.device)
    lse = torch.empty(B, d // 64, L, dtype=torch.float32, device=qkv.device)
    lib = _lib.load()
    ws = torch.empty(lib.mgx_rel_attn_fwd_workspace(L), dtype=torch.uint8, device=qkv.device)
    check(lib.mgx_rel_attn_fwd_nomask(ptr(qkv), ptr(E), ptr(ctx), ptr(lse), ptr(ws), ws.numel(), B, L, L if Lk is None else int(Lk), d,
                                      E.shape[0], stream_ptr()), "mgx_rel_attn_fwd_nomask")
    return ctx


def rel_attn_weights(qkv, E, padbits, lse) -> torch.Tensor:
    """materialised attention weights f32 [B,h,L,L] (eval/debug output of the reference)"""
    _need_cuda(qkv, E, padbits, lse)
    B, L, d3 = qkv.shape
    d = d3 // 3
    w = torch.zeros(B, d // 64, L, L, dtype=torch.float32, device=qkv.device)
    lib = _lib.load()
    ws = torch.empty(lib.mgx_rel_attn_fwd_workspace(L), dtype=torch.uint8, device=qkv.device)
    check(lib.mgx_rel_attn_weights(ptr(qkv), ptr(E), ptr(padbits), ptr(lse), ptr(w), ptr(ws), ws.numel(), B, L, d,
                                   E.shape[0], stream_ptr()), "mgx_rel_attn_weights")
    return w


_SIDE_STREAMS = {}
_CONCURRENT_BWD = __import__("os").environ.get("MGX_CONCURRENT_BWD", "0") == "1"


def _side_streams(dev):
    ss = _SIDE_STREAMS.get(dev)
    if ss is None:
        ss = _SIDE_STREAMS[dev] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return ss


def rel_attn_bwd(qkv, E, padbits, ctx, dctx, lse, dE, parts=15, dqkv=None, workspace=None, concurrent=None) -> torch.Tensor:
    """-> dqkv bf16 [B,L,3d]; dE f32 [M,64] accumulated in place.  parts selects sub-kernels (bench, cross-checks):
    1 pre-pass | 4 dK/dV (stores its dS tiles in the workspace) | 2 dQ from those tiles | 8 dE from those tiles |
    16 dE by recomputation | 32 dQ by recomputation (instead of 2) | 64 dK/dV by the 32-key kernel (instead of 4: cross-check).
    concurrent (opt-in, MGX_CONCURRENT_BWD=1): after dK/dV, the two HBM-bound readers of the dS tiles (dQ, dE) run on two
    streams.  Default is one stream, which keeps per-kernel profiles comparable."""
    _need_cuda(qkv, E, padbits, ctx, dctx, lse, dE)
    B, L, d3 = qkv.shape
    d = d3 // 3
    lib = _lib.load()
    dqkv = torch.empty_like(qkv) if dqkv is None else dqkv
    need = lib.mgx_rel_attn_bwd_workspace(B, L, d)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=qkv.device)
    args = (ptr(qkv), ptr(E), ptr(padbits), ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), ptr(dE), ptr(workspace),
            workspace.numel(), B, L, d, E.shape[0])
    if concurrent is None:
        concurrent = (_CONCURRENT_BWD and parts == 15 and not torch.cuda.is_current_stream_capturing())
    if not concurrent:
        check(lib.mgx_rel_attn_bwd_parts(*args, int(parts), stream_ptr()), "mgx_rel_attn_bwd")
        return dqkv
    main = torch.cuda.current_stream()
    s1, _ = _side_streams(qkv.device)
    check(lib.mgx_rel_attn_bwd_parts(*args, 1 | 4, main.cuda_stream), "mgx_rel_attn_bwd(pre, dKV)")
    ready = torch.cuda.Event()
    ready.record(main)
    s1.wait_event(ready)
    check(lib.mgx_rel_attn_bwd_parts(*args, 8, s1.cuda_stream), "mgx_rel_attn_bwd(dE)")
    check(lib.mgx_rel_attn_bwd_parts(*args, 2, main.cuda_stream), "mgx_rel_attn_bwd(dQ)")
    main.wait_stream(s1)
    return dqkv


def add_ln_fwd(x, res, gamma, beta, eps=1e-6, p_drop=0.0, seed=0):
    _need_cuda(x, res, gamma, beta)
    d = x.shape[-1]
    rows = x.numel() // d
    out = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(_lib.load().mgx_add_ln_fwd(ptr(x), ptr(res), ptr(gamma), ptr(beta), ptr(out), ptr(mean), ptr(rstd), rows,
                                     d, float(eps), float(p_drop), int(seed), stream_ptr()), "mgx_add_ln_fwd")
    return out, mean, rstd


_LN_WS = {}


def add_ln_bwd(dout, x, res, gamma, mean, rstd, dgamma, dbeta, p_drop=0.0, seed=0, dxsum=None):
    """-> (dx, dres); dgamma/dbeta (and dxsum = colsum(dx), if given) are accumulated in place."""
    _need_cuda(dout, x, res, gamma, mean, rstd, dgamma, dbeta, dxsum)
    d = x.shape[-1]
    rows = x.numel() // d
    dres = torch.empty_like(x)
    dx = torch.empty_like(x) if p_drop > 0 else dres
    lib = _lib.load()
    key = (x.device, d)
    ws = _LN_WS.get(key)          # per-device scratch, reused by every call on the (single) compute stream
    need = lib.mgx_add_ln_bwd_workspace(rows, d)
    if ws is None or ws.numel() < need:
        ws = _LN_WS[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
    check(lib.mgx_add_ln_bwd(ptr(dout), ptr(x), ptr(res), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dres),
                             ptr(dgamma), ptr(dbeta), ptr(dxsum), ptr(ws), ws.numel(), rows, d, float(p_drop), int(seed),
                             stream_ptr()), "mgx_add_ln_bwd")
    return dx, dres


def smooth_ce_fwd(logits, target, V, eps_ls, pad):
    """logits bf16 [rows, ld] -> (stats f32[4], argmax int32[rows], row_lse f32[rows])"""
    _need_cuda(logits, target)
    ld = logits.shape[-1]
    rows = logits.numel() // ld
    stats = torch.zeros(4, dtype=torch.float32, device=logits.device)
    argmax = torch.empty(rows, dtype=torch.int32, device=logits.device)
    row_lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    check(_lib.load().mgx_smooth_ce_fwd(ptr(logits), ptr(target), ptr(stats), ptr(argmax), ptr(row_lse), rows, V, ld,
                                        float(eps_ls), int(pad), stream_ptr()), "mgx_smooth_ce_fwd")
    return stats, argmax, row_lse


def smooth_ce_bwd(logits, target, stats, row_lse, V, eps_ls, pad, gscale=1.0, gscale_dev=None):
    """gscale_dev: optional 0-d / 1-element f32 device tensor multiplied into the gradient inside the kernel"""
    _need_cuda(logits, target, stats, row_lse)
    if gscale_dev is not None:
        _need_cuda(gscale_dev)
        if gscale_dev.dtype != torch.float32 or gscale_dev.numel() != 1:
            raise ValueError("gscale_dev must be one float32 element")
    ld = logits.shape[-1]
    rows = logits.numel() // ld
    dlogits = torch.empty_like(logits)
    check(_lib.load().mgx_smooth_ce_bwd(ptr(logits), ptr(target), ptr(stats), ptr(row_lse), ptr(dlogits), rows, V, ld,
                                        float(eps_ls), int(pad), float(gscale), ptr(gscale_dev), stream_ptr()), "mgx_smooth_ce_bwd")
    return dlogits


def adam_step(p, g, m, v, shadow, lr, beta1, beta2, eps, step, gscale=1.0):
    _need_cuda(p, g, m, v, shadow)
    check(_lib.load().mgx_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), p.numel(), float(lr), float(beta1),
                                    float(beta2), float(eps), int(step), float(gscale), stream_ptr()), "mgx_adam_step")


def cast_bf16(p, shadow):
    _need_cuda(p, shadow)
    check(_lib.load().mgx_cast_bf16(ptr(p), ptr(shadow), p.numel(), stream_ptr()), "mgx_cast_bf16")


class FragWeight:
    """a projection weight [N,K] kept in MFMA fragment order (pack_frag; rows zero-padded to a multiple of 32) for the
    decode-size (<= 32 rows) projections: linear_fwd / linear_ln_fwd / decode_embed_linear take it in place of the matrix"""

    def __init__(self, w: torch.Tensor):
        N, K = w.shape
        Np = (N + 31) // 32 * 32
        wp = torch.zeros(Np, K, dtype=BF16, device=w.device)
        wp[:N] = w
        self.data, self.shape = pack_frag(wp), (N, K)


def linear_fwd(a, w, bias, act=0):
    """a bf16 [M,K], w bf16 [N,K] (or a FragWeight, rows <= 32), bias f32 [N] or None -> bf16 [M,N] = act(a @ w.T + bias)"""
    K = a.shape[-1]
    Mrows = a.numel() // K
    N = w.shape[0]
    out = torch.empty(*a.shape[:-1], N, dtype=BF16, device=a.device)
    if isinstance(w, FragWeight):
        _need_cuda(a, w.data, bias)
        check(_lib.load().mgx_skinny_fwd_frag(ptr(a), ptr(w.data), ptr(bias), ptr(out), Mrows, N, K, int(act), stream_ptr()),
              "mgx_skinny_fwd_frag")
        return out
    _need_cuda(a, w, bias)
    check(_lib.load().mgx_linear_fwd(ptr(a), ptr(w), ptr(bias), ptr(out), Mrows, N, K, int(act), stream_ptr()),
          "mgx_linear_fwd")
    return out


def linear_ln_fwd(x, res, gamma, beta, w, bias, act=0, eps=1e-6):
    """decode-size rows (<= 32): z = LayerNorm(x + res), c = act(z @ w^T + bias) in one launch -> (c, z)"""
    N, K = w.shape
    Mrows = x.numel() // K
    c = torch.empty(*x.shape[:-1], N, dtype=BF16, device=x.device)
    z = torch.empty_like(x)
    if isinstance(w, FragWeight):
        _need_cuda(x, res, gamma, beta, w.data, bias)
        check(_lib.load().mgx_linear_ln_fwd_frag(ptr(x), ptr(res), ptr(gamma), ptr(beta), float(eps), ptr(w.data), ptr(bias), ptr(c),
                                                 ptr(z), Mrows, N, K, int(act), stream_ptr()), "mgx_linear_ln_fwd_frag")
        return c, z
    _need_cuda(x, res, gamma, beta, w, bias)
    check(_lib.load().mgx_linear_ln_fwd(ptr(x), ptr(res), ptr(gamma), ptr(beta), float(eps), ptr(w), ptr(bias), ptr(c), ptr(z),
                                        Mrows, N, K, int(act), stream_ptr()), "mgx_linear_ln_fwd")
    return c, z


def linear_dx(dy, w, relu_y=None, addend=None):
    """dy bf16 [..,N], w bf16 [N,K] -> dx bf16 [..,K] = dy @ w (zeroed where relu_y <= 0) (+ addend)"""
    _need_cuda(dy, w, relu_y, addend)
    N, K = w.shape
    Mrows = dy.numel() // N
    dx = torch.empty(*dy.shape[:-1], K, dtype=BF16, device=dy.device)
    if addend is not None and (addend.dtype != BF16 or addend.numel() != dx.numel() or not addend.is_contiguous()):
        raise ValueError("linear_dx: addend must be a contiguous bf16 tensor of dx's shape")
    check(_lib.load().mgx_linear_dx(ptr(dy), ptr(w), ptr(relu_y), ptr(addend), ptr(dx), Mrows, N, K, stream_ptr()),
          "mgx_linear_dx")
    return dx


def linear_dw(dy, x, gw, gb=None):
    """gw f32 [N,K] += dy^T @ x ; gb f32 [N] += colsum(dy).  A weight the ring kernel takes (mgx_linear_dw_grouped_workspace() > 0:
    many rows, tiles mostly full -- the vocabulary projection) goes there as a group of one: partial tiles + fix-up pass instead of
    fp32 atomics"""
    _need_cuda(dy, x, gw, gb)
    N, K = gw.shape
    Mrows = dy.numel() // N
    if dy.is_contiguous() and x.is_contiguous():
        one = (_DwProblem * 1)(_DwProblem(ptr(dy), ptr(x), ptr(gw), ptr(gb), N, K))
        if _lib.load().mgx_linear_dw_grouped_workspace(ctypes.cast(one, ctypes.c_void_p), 1, Mrows):
            return linear_dw_grouped([(dy, x, gw, gb)])
    check(_lib.load().mgx_linear_dw(ptr(dy), ptr(x), ptr(gw), ptr(gb), Mrows, N, K, stream_ptr()), "mgx_linear_dw")


# ---- Event_Melody_RNN training ops (raw launchers; the autograd node lives in melody_rnn.py) ----------------------
def gru_cell_fwd(gi, gh, h_prev, h_next, y):
    """h_next f32 / y bf16 [B,H] <- GRU cell(gi, gh bf16 [B,3H], h_prev f32 [B,H])"""
    _need_cuda(gi, gh, h_prev, h_next, y)
    B, H = h_prev.shape
    check(_lib.load().mgx_gru_cell_fwd(ptr(gi), ptr(gh), ptr(h_prev), ptr(h_next), ptr(y), B, H, stream_ptr()), "mgx_gru_cell_fwd")


def gru_cell_bwd(gi, gh, h_prev, dh_direct, d_rec, dy, dgi, dgh, dh_prev_direct):
    _need_cuda(gi, gh, h_prev, dh_direct, d_rec, dy, dgi, dgh, dh_prev_direct)
    B, H = h_prev.shape
    check(_lib.load().mgx_gru_cell_bwd(ptr(gi), ptr(gh), ptr(h_prev), ptr(dh_direct), ptr(d_rec), ptr(dy), ptr(dgi), ptr(dgh),
                                       ptr(dh_prev_direct), B, H, stream_ptr()), "mgx_gru_cell_bwd")


def pack_frag(w: torch.Tensor) -> torch.Tensor:
    """W [N,K] (N % 32 == 0, K % 16 == 0) -> bf16 copy in MFMA fragment order (mgx.h, fused GRU step): unit
    ((nt*K/16 + ks)*64 + lane) = W[32 nt + lane % 32][16 ks + 8 (lane // 32) .. +7]"""
    N, K = w.shape
    if N % 32 or K % 16:
        raise ValueError("pack_frag: need N % 32 == 0 and K % 16 == 0")
    return w.to(BF16).view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


def gru_step_fwd(gi, h_prev_bf, h_prev, whh, bhh, h_next, y, gh_out):
    """one fused time step (whh = pack_frag(W_hh)): gh_out bf16 [B,3H] = h_prev_bf @ W_hh.T + bhh, then h_next f32 / y bf16 [B,H] = cell(gi, gh, h_prev)"""
    _need_cuda(gi, h_prev_bf, h_prev, whh, bhh, h_next, y, gh_out)
    B, H = h_prev.shape
    check(_lib.load().mgx_gru_step_fwd(ptr(gi), ptr(h_prev_bf), ptr(h_prev), ptr(whh), ptr(bhh), ptr(h_next), ptr(y), ptr(gh_out),
                                       B, H, stream_ptr()), "mgx_gru_step_fwd")


def gru_step_x_fwd(x, wih, bih, h_prev_bf, h_prev, whh, bhh, h_next, y):
    """sampling step of one GRU layer in one launch (wih, whh = pack_frag of the weights): both projections + the cell;
    (h_next, y) must not alias (h_prev, h_prev_bf)"""
    _need_cuda(x, wih, bih, h_prev_bf, h_prev, whh, bhh, h_next, y)
    B, H = h_prev.shape
    check(_lib.load().mgx_gru_step_x_fwd(ptr(x), ptr(wih), ptr(bih), wih.shape[1], ptr(h_prev_bf), ptr(h_prev), ptr(whh), ptr(bhh),
                                         ptr(h_next), ptr(y), B, H, stream_ptr()), "mgx_gru_step_x_fwd")


def gru_step_bwd(gi, gh, h_prev, dh_direct, dgh_next, whh_t, dy, dgi, dgh, dh_out, final=False):
    """one fused backward step: d_rec = dgh_next @ W_hh (whh_t = pack_frag(W_hh.T)), then the cell backward (see mgx.h)"""
    _need_cuda(gi, gh, h_prev, dh_direct, dgh_next, whh_t, dy, dgi, dgh, dh_out)
    B, H = dh_out.shape
    check(_lib.load().mgx_gru_step_bwd(ptr(gi), ptr(gh), ptr(h_prev), ptr(dh_direct), ptr(dgh_next), ptr(whh_t), ptr(dy), ptr(dgi),
                                       ptr(dgh), ptr(dh_out), B, H, 1 if final else 0, stream_ptr()), "mgx_gru_step_bwd")


def dropout_bf16(x, p_drop, seed):
    """stateless inverted dropout; the same call on a gradient is the backward"""
    _need_cuda(x)
    if p_drop <= 0:
        return x
    out = torch.empty_like(x)
    check(_lib.load().mgx_dropout_bf16(ptr(x), ptr(out), x.numel(), float(p_drop), int(seed), stream_ptr()), "mgx_dropout_bf16")
    return out


def scatter_add_rows(idx, src, dst):
    """dst f32 [V,cols][idx[r]] += src bf16 [n,ld][r, :cols]"""
    _need_cuda(idx, src, dst)
    V, cols = dst.shape
    n, ld = src.shape
    check(_lib.load().mgx_scatter_add_rows(ptr(idx), ptr(src), ptr(dst), n, ld, cols, V, stream_ptr()), "mgx_scatter_add_rows")


_DW_WS = {}


class _DwProblem(ctypes.Structure):          # mirrors mgx_dw_problem (include/mgx.h)
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("gW", ctypes.c_void_p), ("gb", ctypes.c_void_p),
                ("N", ctypes.c_int), ("K", ctypes.c_int)]


def linear_dw_grouped(problems):
    """problems: list of (dy bf16 [..,N], x bf16 [..,K], gw f32 [N,K], gb f32 [N] or None) sharing the row count:
    all weight (and bias) gradients in one launch."""
    arr = (_DwProblem * len(problems))()
    Mrows = None
    for i, (dy, x, gw, gb) in enumerate(problems):
        _need_cuda(dy, x, gw, gb)
        N, K = gw.shape
        rows = dy.numel() // N
        if Mrows is None:
            Mrows = rows
        if rows != Mrows or x.numel() != rows * K or not (dy.is_contiguous() and x.is_contiguous()):
            raise ValueError("linear_dw_grouped: all problems need contiguous dy [M,N], x [M,K] with the same M")
        arr[i] = _DwProblem(ptr(dy), ptr(x), ptr(gw), ptr(gb), N, K)
    lib = _lib.load()
    parr = ctypes.cast(arr, ctypes.c_void_p)
    need = lib.mgx_linear_dw_grouped_workspace(parr, len(problems), Mrows)
    dev = problems[0][0].device
    # scratch for the fp32 partial tiles of the M-splits, one per (device, stream): two calls on different streams must not
    # share partial tiles, and a buffer is only ever replaced by a larger one for the stream that asked (the caching allocator
    # keeps the old block alive until that stream's work on it is done; never while a graph is being captured)
    key = (dev, stream_ptr())
    ws = _DW_WS.get(key)
    if need and (ws is None or ws.numel() < need):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("linear_dw_grouped: the workspace must exist before graph capture (run one step eagerly first)")
        ws = _DW_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    check(lib.mgx_linear_dw_grouped(parr, len(problems), Mrows, ptr(ws) if need else None, need, stream_ptr()),
          "mgx_linear_dw_grouped")


# ---- decode path (no autograd) ----------------------------------------------------------------------
def decode_embed(tok, table, pe, pos_dev, out):
    _need_cuda(tok, table, pe, pos_dev, out)
    V, d = table.shape
    check(_lib.load().mgx_decode_embed(ptr(tok), ptr(table), ptr(pe), ptr(pos_dev), ptr(out), tok.numel(), d, V,
                                       stream_ptr()), "mgx_decode_embed")
    return out


def decode_embed_linear(tok, table, pe, pos_dev, w, bias, hout):
    """h = table[tok]*sqrt(d) + pe[t] (written to ``hout`` bf16 [B,d]) and c bf16 [B,N] = h w^T + bias in one launch"""
    V, d = table.shape
    N = w.shape[0]
    c = torch.empty(tok.numel(), N, dtype=torch.bfloat16, device=hout.device)
    if isinstance(w, FragWeight):
        _need_cuda(tok, table, pe, pos_dev, w.data, bias, hout)
        check(_lib.load().mgx_decode_embed_linear_frag(ptr(tok), ptr(table), ptr(pe), ptr(pos_dev), ptr(w.data), ptr(bias), ptr(c),
                                                       ptr(hout), tok.numel(), N, d, V, stream_ptr()), "mgx_decode_embed_linear_frag")
        return c, hout
    _need_cuda(tok, table, pe, pos_dev, w, bias, hout)
    check(_lib.load().mgx_decode_embed_linear(ptr(tok), ptr(table), ptr(pe), ptr(pos_dev), ptr(w), ptr(bias), ptr(c), ptr(hout),
                                              tok.numel(), N, d, V, stream_ptr()), "mgx_decode_embed_linear")
    return c, hout


def rel_attn_decode_splits(B, Lmax, d) -> int:
    return int(_lib.load().mgx_rel_attn_decode_splits(B, Lmax, d))


def rel_attn_decode_workspace(B, Lmax, d, device):
    """scratch for the split-K partials of mgx_rel_attn_decode (None when the cache is short enough for one workgroup per
    (b,h)); allocate once per generation: the decode step is graph-captured, so the buffer must outlive the graph"""
    n = _lib.load().mgx_rel_attn_decode_workspace(B, Lmax, d)
    return torch.empty(n, dtype=torch.uint8, device=device) if n else None


def rel_attn_decode(qkv_new, kcache, vcache, E, pos_dev, ctx, workspace=None):
    """kcache / vcache bf16 [B, h, Lmax, 64] (head-major: workgroup (b, h) streams one contiguous run of rows)"""
    _need_cuda(qkv_new, kcache, vcache, E, pos_dev, ctx, workspace)
    B, heads, Lmax, dh = kcache.shape
    if dh != 64 or vcache.shape != kcache.shape:
        raise ValueError("rel_attn_decode: caches must be bf16 [B, h, Lmax, 64]")
    d = heads * 64
    check(_lib.load().mgx_rel_attn_decode(ptr(qkv_new), ptr(kcache), ptr(vcache), ptr(E), ptr(pos_dev), ptr(ctx), ptr(workspace),
                                          0 if workspace is None else workspace.numel(), B, Lmax, d, E.shape[0], stream_ptr()),
          "mgx_rel_attn_decode")
    return ctx


def sample_topk_topp(logits, V, pos_dev, next_tok, out_tokens=None, probs_out=None, temperature=1.0, top_k=0, top_p=1.0,
                     seed=0, advance=True, allow_table=None, row0=0):
    """allow_table: optional int32/uint32 [V, ceil(V/32)] grammar mask on the device (bit v of row t: v may follow t);
    row0: index of the first row in the whole batch when the tensors are a sub-batch's rows (the draw is by global row)"""
    _need_cuda(logits, pos_dev, next_tok, out_tokens, probs_out, allow_table)
    ld = logits.shape[-1]
    B = logits.numel() // ld
    if allow_table is not None and (allow_table.dim() != 2 or allow_table.shape[0] != V
                                    or allow_table.shape[1] != (V + 31) // 32 or allow_table.element_size() != 4
                                    or not allow_table.is_contiguous()):
        raise ValueError("allow_table must be a contiguous 32-bit integer tensor of shape [V, ceil(V/32)]")
    check(_lib.load().mgx_sample_topk_topp_rows(ptr(logits), int(V), ld, float(temperature), int(top_k), float(top_p), int(seed),
                                                ptr(pos_dev), ptr(next_tok), ptr(out_tokens),
                                                0 if out_tokens is None else out_tokens.shape[-1], ptr(probs_out), B, int(row0),
                                                1 if advance else 0, ptr(allow_table), stream_ptr()), "mgx_sample_topk_topp")
    return next_tok


# --------------------------------------------------------------------------------------------------
# autograd glue
#
# Master (fp32) parameters are passed to each Function only to anchor the autograd graph: their
# gradients are NOT returned (None) but accumulated by the kernels straight into the fp32 views of
# the model's flat gradient buffer (``g*`` arguments).  ``done`` is an optional callable invoked at
# the end of backward (the data-parallel bucket hook, see dp.py).
# --------------------------------------------------------------------------------------------------
class _EmbedPE(torch.autograd.Function):
    """K1: dropout(emb[x]*sqrt(d) + PE)                       layers.py:226-229"""

    @staticmethod
    def forward(ctx, tok, table, pe, p_drop, seed, gtable, done):
        ctx.save_for_backward(tok)
        ctx.cfg = (p_drop, seed, gtable, done)
        return embed_pe_fwd(tok, table, pe, p_drop, seed)

    @staticmethod
    def backward(ctx, dout):
        (tok,) = ctx.saved_tensors
        p_drop, seed, gtable, done = ctx.cfg
        embed_bwd(tok, dout.contiguous(), gtable, p_drop, seed)
        if done is not None:
            done()
        join_side_stream(tok.device)      # last node of the backward: every gradient is complete in this stream's order
        return None, None, None, None, None, None, None


class _Linear(torch.autograd.Function):
    """K2/K5/K7/K8: y = act(x @ W^T + b) and its backward, all libmgx MFMA kernels:
    forward NT GEMM with fused bias/ReLU; dx = dy @ W (NN, transposed LDS reads); dW += dy^T @ x and
    db += colsum(dy) accumulate in fp32 straight into the flat gradient buffer (TN, split-M atomics).
    ``x_is_relu``: x is the output of a fused-ReLU layer, so dx is masked by (x > 0) in the dX
    epilogue -- that producer (act=1) then receives an already-masked gradient."""

    @staticmethod
    def forward(ctx, x, w_master, w_shadow, bias, act, gw, gb, done, x_is_relu):
        y = linear_fwd(x, w_shadow, bias, act)
        ctx.save_for_backward(x, w_shadow)
        ctx.cfg = (gw, gb, done, x_is_relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_shadow = ctx.saved_tensors
        gw, gb, done, x_is_relu = ctx.cfg
        dy = dy.contiguous()
        dx = linear_dx(dy, w_shadow, x if x_is_relu else None)
        linear_dw(dy, x, gw, gb)
        if done is not None:
            done()
        return dx, None, None, None, None, None, None, None, None


class LayerParams:
    """Views one encoder layer's kernels need (bf16 shadow weights, fp32 biases / LayerNorm parameters and
    the fp32 gradient slots they accumulate into); built once per model by network.MusicTransformer."""
    __slots__ = ("wqkv", "bqkv", "gqkv", "gbqkv", "E", "gE", "wfc", "bfc", "gwfc", "gbfc", "g1", "b1", "gg1", "gb1",
                 "wpre", "bpre", "gwpre", "gbpre", "wsuf", "bsuf", "gwsuf", "gbsuf", "g2", "b2", "gg2", "gb2")


class _EncoderLayer(torch.autograd.Function):
    """One whole post-LN block (layers.py:152-161) as a single autograd node: 7 kernels forward, 14 backward.
    Doing the block in one node lets the gradient that arrives over each residual connection join the
    branch's input gradient inside the dX GEMM epilogue (``addend``) instead of a separate elementwise pass,
    and frees each intermediate as soon as its last consumer has run."""

    @staticmethod
    def forward(ctx, h, lp, padbits, p_drop, seed, done, wsink):
        qkv = linear_fwd(h, lp.wqkv, lp.bqkv, 0)
        att, lse = rel_attn_fwd(qkv, lp.E, padbits)
        if wsink is not None:
            wsink.append(rel_attn_weights(qkv, lp.E, padbits, lse))
        a = linear_fwd(att, lp.wfc, lp.bfc, 0)
        o1, mean1, rstd1 = add_ln_fwd(a, h, lp.g1, lp.b1, 1e-6, p_drop, seed + 1)
        f1 = linear_fwd(o1, lp.wpre, lp.bpre, 1)
        f2 = linear_fwd(f1, lp.wsuf, lp.bsuf, 0)
        out, mean2, rstd2 = add_ln_fwd(f2, o1, lp.g2, lp.b2, 1e-6, p_drop, seed + 2)
        ctx.save_for_backward(h, qkv, att, lse, a, o1, f1, f2, mean1, rstd1, mean2, rstd2)
        ctx.cfg = (lp, padbits, p_drop, seed, done)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, qkv, att, lse, a, o1, f1, f2, mean1, rstd1, mean2, rstd2 = ctx.saved_tensors
        lp, padbits, p_drop, seed, done = ctx.cfg
        # LN2 <- FFN_suf <- ReLU <- FFN_pre, residual o1
        df2, dres2 = add_ln_bwd(dout.contiguous(), f2, o1, lp.g2, mean2, rstd2, lp.gg2, lp.gb2, p_drop, seed + 2, lp.gbsuf)
        df1 = linear_dx(df2, lp.wsuf, f1)
        do1 = linear_dx(df1, lp.wpre, None, dres2)
        del dres2
        # LN1 <- fc <- attention <- QKV, residual h
        da, dres1 = add_ln_bwd(do1, a, h, lp.g1, mean1, rstd1, lp.gg1, lp.gb1, p_drop, seed + 1, lp.gbfc)
        del do1
        datt = linear_dx(da, lp.wfc)
        plan = None if torch.cuda.is_current_stream_capturing() else stream_plan(h.device)
        if plan is None or plan.side is None:
            dqkv = rel_attn_bwd(qkv, lp.E, padbits, att, datt, lse, lp.gE)
            del datt
            dh = linear_dx(dqkv, lp.wqkv, None, dres1)
            # the block's four weight gradients (and the two bias gradients that are not LayerNorm by-products) in one launch
            linear_dw_grouped([(dqkv, h, lp.gqkv, lp.gbqkv), (da, att, lp.gwfc, None), (df1, o1, lp.gwpre, lp.gbpre),
                               (df2, f1, lp.gwsuf, None)])
            if done is not None:
                done()
            return dh, None, None, None, None, None, None
        # Two streams (configure_streams): dE and the weight gradients feed only the optimiser (and the bucket's all-reduce), and
        # both run at the HBM rate -- they go to the CU-masked side stream, behind an event recorded once dK/dV (the dS tiles) and
        # dQ (dqkv complete) are issued, while this stream carries on with the dX of the QKV projection and the next block's
        # backward, whose heavy kernels are MFMA-bound.  The bucket callback runs in the side stream's context: the all-reduce
        # it issues is ordered behind the gradients written there (everything this stream wrote into the bucket -- the
        # LayerNorm and bias column sums -- precedes the event).
        main, side = torch.cuda.current_stream(), plan.side.stream
        B, L, _ = qkv.shape
        ws = torch.empty(_lib.load().mgx_rel_attn_bwd_workspace(B, L, qkv.shape[2] // 3), dtype=torch.uint8, device=qkv.device)
        de_side, dw_side = bool(plan.work & 1), bool(plan.work & 2)
        dqkv = rel_attn_bwd(qkv, lp.E, padbits, att, datt, lse, lp.gE, parts=(1 | 4 | 2) if de_side else 15, workspace=ws)
        ready = torch.cuda.Event()
        ready.record(main)
        dh = linear_dx(dqkv, lp.wqkv, None, dres1)
        group = [(dqkv, h, lp.gqkv, lp.gbqkv), (da, att, lp.gwfc, None), (df1, o1, lp.gwpre, lp.gbpre), (df2, f1, lp.gwsuf, None)]
        if not dw_side:
            linear_dw_grouped(group)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            if de_side:
                rel_attn_bwd(qkv, lp.E, padbits, att, datt, lse, lp.gE, parts=8, dqkv=dqkv, workspace=ws)
            if dw_side:
                linear_dw_grouped(group)
            if done is not None:
                if not dw_side:                            # the weight gradients were issued on the other stream, after `ready`
                    again = torch.cuda.Event()
                    again.record(main)
                    side.wait_event(again)
                done()
            plan.last = torch.cuda.Event()
            plan.last.record(side)
        # allocated on this stream, read on the other: the caching allocator must not hand the blocks out again before the side
        # stream is through with them
        for t in (qkv, att, datt, lse, ws, dqkv, h, da, df1, o1, df2, f1, padbits):
            if t is not None:
                t.record_stream(side)
        return dh, None, None, None, None, None, None


def encoder_layer(h, lp, padbits, p_drop, seed, done=None, wsink=None):
    return _EncoderLayer.apply(h, lp, padbits, p_drop, seed, done, wsink)


class _SmoothCE(torch.autograd.Function):
    """K9+K10: smoothed CE (mean over non-pad) with accuracy/argmax side outputs.   criterion.py:43-67"""

    @staticmethod
    def forward(ctx, logits, target, V, eps_ls, pad):
        stats, argmax, row_lse = smooth_ce_fwd(logits, target, V, eps_ls, pad)
        ctx.save_for_backward(logits, target, stats, row_lse)
        ctx.cfg = (V, eps_ls, pad)
        ctx.mark_non_differentiable(stats, argmax)
        loss = stats[0] / stats[1]
        return loss, stats, argmax

    @staticmethod
    def backward(ctx, gloss, _gs, _ga):
        logits, target, stats, row_lse = ctx.saved_tensors
        V, eps_ls, pad = ctx.cfg
        # gloss is a 0-d device tensor: the kernel reads it on the device (no host sync, no second pass over dlogits)
        dl = smooth_ce_bwd(logits, target, stats, row_lse, V, eps_ls, pad, 1.0, gloss.detach().to(torch.float32).contiguous())
        return dl, None, None, None, None


def embed_pe(tok, table, pe, p_drop=0.0, seed=0, gtable=None, done=None):
    return _EmbedPE.apply(tok, table, pe, float(p_drop), int(seed), gtable, done)


# --------------------------------------------------------------------------------------------------
# Stand-alone autograd nodes: the layer-level call surface of the reference (layers.py:64-109,152-161,
# 223-233 -- ``rga([q,k,v], mask)``, ``EncoderLayer(x, mask)``, ``Encoder(x, mask)``).  Unlike the nodes
# above they take ordinary fp32 Parameters, round them to bf16 per call and RETURN their gradients, so a
# layer works on its own with any torch optimizer.  Same kernels; the model's training path keeps using the
# flat-buffer nodes above (no per-call rounding, gradients accumulated in place).
# --------------------------------------------------------------------------------------------------
class _LinearStd(torch.autograd.Function):
    """y = act(x @ W^T + b) with returned gradients.  A reduction length that is no multiple of the GEMMs' granule of 64 (the FFN's
    second projection at d_model = 64 * odd: K = d / 2) is zero-padded here, per call; the model's flat-buffer path pads once,
    inside its buffers (network._flat_order)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        w16 = weight.detach().to(BF16).contiguous()
        K = w16.shape[1]
        Kp = (K + 63) // 64 * 64
        if Kp != K:
            w16 = torch.nn.functional.pad(w16, (0, Kp - K))
            x = torch.nn.functional.pad(x, (0, Kp - K))
        y = linear_fwd(x.contiguous(), w16, None if bias is None else bias.detach().float().contiguous(), act)
        ctx.save_for_backward(x, w16, y if act else None)
        ctx.meta = (weight.dtype, bias is not None, bias.dtype if bias is not None else None, K)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w16, y = ctx.saved_tensors
        wdt, has_b, bdt, K = ctx.meta
        dy = dy.contiguous()
        gw = torch.zeros(w16.shape, dtype=torch.float32, device=dy.device)
        gb = torch.zeros(w16.shape[0], dtype=torch.float32, device=dy.device) if has_b else None
        if y is not None:                           # ReLU output: mask the incoming gradient once, for dx and dW alike
            dy = torch.where(y > 0, dy, torch.zeros_like(dy))
        dx = linear_dx(dy, w16)
        linear_dw(dy, x, gw, gb)
        if w16.shape[1] != K:
            dx, gw = dx[..., :K], gw[:, :K]
        return dx, gw.to(wdt), (gb.to(bdt) if has_b else None), None


class _RelAttnStd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, E, padbits):
        e16 = E.detach().to(BF16).contiguous()
        att, lse = rel_attn_fwd(qkv, e16, padbits)
        ctx.save_for_backward(qkv, e16, att, lse)
        ctx.padbits, ctx.edt = padbits, E.dtype
        ctx.mark_non_differentiable(lse)
        return att, lse

    @staticmethod
    def backward(ctx, datt, _dlse):
        qkv, e16, att, lse = ctx.saved_tensors
        dE = torch.zeros(e16.shape, dtype=torch.float32, device=qkv.device)
        dqkv = rel_attn_bwd(qkv, e16, ctx.padbits, att, datt.contiguous(), lse, dE)
        return dqkv, dE.to(ctx.edt), None


class _AddLNStd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p_drop, seed):
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        out, mean, rstd = add_ln_fwd(x, res, g32, b32, eps, p_drop, seed)
        ctx.save_for_backward(x, res, g32, mean, rstd)
        ctx.meta = (p_drop, seed, gamma.dtype, beta.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, res, g32, mean, rstd = ctx.saved_tensors
        p_drop, seed, gdt, bdt = ctx.meta
        dg, db = torch.zeros_like(g32), torch.zeros_like(g32)
        dx, dres = add_ln_bwd(dout.contiguous(), x, res, g32, mean, rstd, dg, db, p_drop, seed)
        return dx, dres, dg.to(gdt), db.to(bdt), None, None, None


def linear_std(x, weight, bias=None, act=0):
    return _LinearStd.apply(x, weight, bias, int(act))


def rel_attn_std(qkv, E, padbits):
    return _RelAttnStd.apply(qkv, E, padbits)


def add_ln_std(x, res, gamma, beta, eps=1e-6, p_drop=0.0, seed=0):
    return _AddLNStd.apply(x, res, gamma, beta, float(eps), float(p_drop), int(seed))


def linear(x, w_master, w_shadow, bias, act, gw, gb, done=None, x_is_relu=False):
    return _Linear.apply(x, w_master, w_shadow, bias, int(act), gw, gb, done, bool(x_is_relu))


def smooth_ce(logits, target, V, eps_ls, pad):
    """returns (loss scalar tensor, stats f32[4], argmax int32[rows])"""
    return _SmoothCE.apply(logits, target, int(V), float(eps_ls), int(pad))
