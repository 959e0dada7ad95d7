"""Parity of the 64-keys-per-wave dK/dV kernel with the generated asm main loop (csrc/rel_attn_dkv64.hip, parts bit 2 where
L % 128 == 0) with the 32-key kernel (parts bit 6): dk, dv and every stored dS tile must be bit-identical (same arithmetic per
tile, same accumulation order).  GPU box:   python tools/check_dkv64.py     (also localises differing tiles on a small case)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops
dev = torch.device("cuda")
for (B, L, d) in ((1, 256, 64), (2, 256, 128), (1, 1024, 64), (3, 512, 192), (2, 2048, 128), (1, 640, 64), (1, 896, 64)):
    g = torch.Generator().manual_seed(L)
    qkv = (torch.randn(B, L, 3 * d, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    E = (torch.randn(L, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dctx = torch.randn(B, L, d, generator=g).to(torch.bfloat16).to(dev)
    tok = torch.zeros(B, L, dtype=torch.int32); tok[0, L - 37:] = 5
    for bits in (None, ops.pad_bitmap(tok.to(dev), 5)):
        ctx, lse = ops.rel_attn_fwd(qkv, E, bits)
        n = ops._lib.load().mgx_rel_attn_bwd_workspace(B, L, d)
        ws1 = torch.zeros(n, dtype=torch.uint8, device=dev); ws2 = torch.zeros(n, dtype=torch.uint8, device=dev)
        dE = torch.zeros(L, 64, device=dev)
        dq1 = torch.zeros_like(qkv); dq2 = torch.zeros_like(qkv)
        ops.rel_attn_bwd(qkv, E, bits, ctx, dctx, lse, dE, 1 | 64, dq1, ws1)      # 32-key kernel
        ops.rel_attn_bwd(qkv, E, bits, ctx, dctx, lse, dE, 1 | 4, dq2, ws2)       # 64-key kernel, asm main loop
        torch.cuda.synchronize()
        same_kv = torch.equal(dq1[..., d:], dq2[..., d:])
        same_ws = torch.equal(ws1, ws2)
        print(B, L, d, "pad" if bits is not None else "nopad", "dk/dv equal:", same_kv, " dS workspace equal:", same_ws,
              " max|diff|", (dq1[..., d:].float() - dq2[..., d:].float()).abs().max().item())

# localise: which dS tiles differ (B=1, L=256, d=64: 8x8 tile grid, causal half packed by rows)
B, L, d = 1, 512, 64
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(B, L, 3 * d, generator=g) * 0.7).to(torch.bfloat16).to(dev)
E = (torch.randn(L, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
dctx = torch.randn(B, L, d, generator=g).to(torch.bfloat16).to(dev)
ctx, lse = ops.rel_attn_fwd(qkv, E, None)
n = ops._lib.load().mgx_rel_attn_bwd_workspace(B, L, d)
ws1 = torch.zeros(n, dtype=torch.uint8, device=dev); ws2 = torch.zeros(n, dtype=torch.uint8, device=dev)
dE = torch.zeros(L, 64, device=dev); dq1 = torch.zeros_like(qkv); dq2 = torch.zeros_like(qkv)
ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 1 | 64, dq1, ws1)
ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 1 | 4, dq2, ws2)
torch.cuda.synchronize()
nt = L // 32
off = n - nt * (nt + 1) // 2 * 2048
t1 = ws1[off:].view(-1, 2048).cpu(); t2 = ws2[off:].view(-1, 2048).cpu()
bad = []
for I in range(nt):
    for J in range(I + 1):
        k = I * (I + 1) // 2 + J
        if not torch.equal(t1[k], t2[k]): bad.append((I, J))
print("dS tiles that differ (I, J):", bad)
dk1, dk2 = dq1[0, :, d:2 * d].float().cpu(), dq2[0, :, d:2 * d].float().cpu()
dv1, dv2 = dq1[0, :, 2 * d:].float().cpu(), dq2[0, :, 2 * d:].float().cpu()
print("dk rows (key tiles) that differ:", [j for j in range(nt) if not torch.equal(dk1[32 * j:32 * j + 32], dk2[32 * j:32 * j + 32])])
print("dv rows (key tiles) that differ:", [j for j in range(nt) if not torch.equal(dv1[32 * j:32 * j + 32], dv2[32 * j:32 * j + 32])])
