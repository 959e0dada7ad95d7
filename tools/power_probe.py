"""Same kernels, same instruction streams, random against all-zero operands: how much of the attention kernels' time is the clock the chip
holds under the operands' switching activity (GPU box, repo root): python tools/power_probe.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from musicgeneration_amd import ops
dev = torch.device("cuda"); B, L, d = 64, 2048, 512
g = torch.Generator().manual_seed(7)
def run(scale):
    qkv = (torch.randn(B, L, 3 * d, generator=g) * 0.7 * scale).to(torch.bfloat16).to(dev)
    E = (torch.randn(L, 64, generator=g) * 0.5 * scale).to(torch.bfloat16).to(dev)
    dctx = (torch.randn(B, L, d, generator=g) * scale).to(torch.bfloat16).to(dev)
    dE = torch.zeros(L, 64, device=dev)
    ctx, lse = ops.rel_attn_fwd(qkv, E, None)
    dqkv = torch.zeros_like(qkv)
    ws = torch.empty(ops._lib.load().mgx_rel_attn_bwd_workspace(B, L, d), dtype=torch.uint8, device=dev)
    out = {}
    for name, parts in (("fwd", None), ("dkv", 4), ("dq_lite", 2), ("de_tiles", 8)):
        fn = (lambda: ops.rel_attn_fwd(qkv, E, None)) if parts is None else (lambda: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, parts, dqkv, ws))
        if parts == 4: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 1 | 4, dqkv, ws)
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 10
    return out
for sc in (1.0, 0.0, 1.0, 0.0):
    r = run(sc)
    print(f"operand scale {sc}: " + "  ".join(f"{k} {v:.3f} ms" for k, v in r.items()))
