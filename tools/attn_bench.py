"""Micro-benchmark of the attention kernels at the cfg2 shape (for rocprofv3 / A-B timing)."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8); ap.add_argument("--L", type=int, default=2048)
ap.add_argument("--d", type=int, default=512); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--parts", type=int, default=63, help="bit0 fwd, bit1 pre-pass, bit2 dq (from tiles / recompute), bit3 dkv, bit4 de (from tiles), bit5 de (recompute), bit6 whole backward")
ap.add_argument("--rounds", type=int, default=1)
ap.add_argument("--fwd-variants", action="store_true", help="also time the two experimental forwards (needs MGX_LIB_PATH=...libmgx_exp.so)")
a = ap.parse_args()
dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
qkv = (torch.randn(a.B, a.L, 3 * a.d, generator=g) * 0.7).to(torch.bfloat16).to(dev)
E = (torch.randn(a.L, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
dctx = torch.randn(a.B, a.L, a.d, generator=g).to(torch.bfloat16).to(dev)
dE = torch.zeros(a.L, 64, device=dev)
ctx, lse = ops.rel_attn_fwd(qkv, E, None)
dqkv = torch.empty_like(qkv); ws = torch.empty(ops._lib.load().mgx_rel_attn_bwd_workspace(*qkv.shape[:2], qkv.shape[2] // 3), dtype=torch.uint8, device=dev)
ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 15, dqkv, ws)
torch.cuda.synchronize()
unit = a.B * a.L * a.L * a.d
def with_env(k, v, fn):
    def g():
        old = os.environ.get(k); os.environ[k] = v
        try: fn()
        finally:
            if old is None: os.environ.pop(k, None)
            else: os.environ[k] = old
    return g
def timed(fn, units, name):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print(f"{name:8s} {ms:8.3f} ms   executed {units*unit/ms/1e9:8.1f} TF/s ({units} units)")
if a.parts & 1:
    for _ in range(a.rounds):          # interleaved rounds in one process (A/B)
        timed(with_env("MGX_ATTN_FWD64", "0", lambda: ops.rel_attn_fwd(qkv, E, None)), 3, "fwd32")
        if not a.fwd_variants: continue
        timed(with_env("MGX_ATTN_FWD64", "3", lambda: ops.rel_attn_fwd(qkv, E, None)), 3, "fwd64asm")
        timed(with_env("MGX_ATTN_FWD64", "1", lambda: ops.rel_attn_fwd(qkv, E, None)), 3, "fwd64")
        timed(with_env("MGX_ATTN_FWD64", "2", lambda: ops.rel_attn_fwd(qkv, E, None)), 3, "fwdpp")
        timed(with_env("MGX_ATTN_PP_RIGID", "1", with_env("MGX_ATTN_FWD64", "2", lambda: ops.rel_attn_fwd(qkv, E, None))), 3, "fwdpp_r")
if a.parts & 2: timed(lambda: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 1, dqkv, ws), 0, "pre")
if a.parts & 4:
    for _ in range(a.rounds):
        timed(lambda: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 2, dqkv, ws), 2, "dq_lite")     # from the stored dS tiles
        timed(lambda: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 32, dqkv, ws), 5, "dq_rec")     # recompute (cross-check kernel)
if a.parts & 8:
    for _ in range(a.rounds):
        timed(lambda: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 4, dqkv, ws), 5.5 if a.L % 128 == 0 else 6, "dkv")     # 64-key asm kernel where L % 128 == 0
        timed(lambda: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 64, dqkv, ws), 6, "dkv32")                              # the 32-key kernel
if a.parts & 16: timed(lambda: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 8, dqkv, ws), 1, "de_tiles")
if a.parts & 32: timed(lambda: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 16, dqkv, ws), 6, "de_rec")
if a.parts & 64:
    for _ in range(a.rounds):
        timed(lambda: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 15, dqkv, ws), 9, "bwd")       # the whole backward as the step calls it
