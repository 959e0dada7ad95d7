"""debug aid: dump a dS tile of the 64-key asm kernel next to the 32-key kernel's"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops
dev = torch.device("cuda")
B, L, d = 1, 256, 64
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
def tile(t, I, J):
    raw = t[I * (I + 1) // 2 + J].view(torch.bfloat16).float().view(2, 64, 8)     # [ss][lane][k]
    out = torch.zeros(32, 32)
    for ss in range(2):
        for lane in range(64):
            hh, j = lane >> 5, lane & 31
            for k in range(8):
                r = 8 * ss + k
                i = (r & 3) + 8 * (r >> 2) + 4 * hh
                out[i, j] = raw[ss, lane, k]
    return out
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
for (I, J) in ((4, 0), (4, 1), (5, 0), (7, 3)):
    a, b = tile(t1, I, J), tile(t2, I, J)
    print(f"tile ({I},{J}): ref absmax {a.abs().max():.4f}  asm absmax {b.abs().max():.4f}  maxdiff {(a-b).abs().max():.4f}  nan {torch.isnan(b).sum().item()}")
    print("ref[0:6,0:10]\n", a[:6, :10]); print("asm[0:6,0:10]\n", b[:6, :10])
    # is asm a permutation / shift of ref?
    eq = (a == b).float().mean().item(); print("fraction equal", eq)
dk1, dk2 = dq1[0, :, d:2 * d].float().cpu(), dq2[0, :, d:2 * d].float().cpu()
print("dk ref[0:3,0:8]", dk1[:3, :8], "\ndk asm", dk2[:3, :8])
