"""In-kernel cycle shares of the 64-row forward kernel with the generated asm sweep (diagnostic build, never the product):
    python -m musicgeneration_amd._build --variant fwd64stamp --experiments -DMGX_FWD64_STAMP        (here, cross-compiles)
    MGX_ATTN_FWD64=3 MGX_LIB_PATH=musicgeneration_amd/libmgx_fwd64stamp.so python tools/experiments/fwd64_stamp.py [--B 64]   (GPU box)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from musicgeneration_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64); ap.add_argument("--L", type=int, default=2048); ap.add_argument("--d", type=int, default=512)
a = ap.parse_args()
assert "stamp" in os.environ.get("MGX_LIB_PATH", "")
dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
qkv = (torch.randn(a.B, a.L, 3 * a.d, generator=g) * 0.7).to(torch.bfloat16).to(dev)
E = (torch.randn(a.L, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
for _ in range(3): ctx, lse = ops.rel_attn_fwd(qkv, E, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx, lse = ops.rel_attn_fwd(qkv, E, None); e1.record(); torch.cuda.synchronize()
print(f"stamped fwd64 kernel: {e0.elapsed_time(e1):.3f} ms")
h = a.d // 64
rec = lse.view(a.B, h, a.L // 64, 64)[..., :16].reshape(-1, 16).cpu()      # one record per wave (64 rows)
clk = (rec[:, 11].sum() / rec[:, 12].sum()).item()
print(f"clock during the kernel: {clk * 100:.0f} MHz")
print("qb w | main iters  masked iters | asm block   total | main body: ->8  ->16  ->24  sum | masked body: ->8 ->16 ->24 sum | asm outside the bodies")
tot = torch.zeros(5)
for qb in range(a.L // 128 - 1, -1, -1):
    for w in range(2):
        m = rec[(rec[:, 8] == qb) & (rec[:, 9] == w)].mean(0)
        nm, nk = max(m[7].item(), 1), max(m[6].item(), 1)
        mb = [m[k].item() / nm for k in range(3)]; kb = [m[3 + k].item() / nk for k in range(3)]
        inb = m[:6].sum().item()
        if w == 0: tot += torch.tensor([m[:3].sum().item(), m[3:6].sum().item(), m[10].item() - inb, m[11].item() - m[10].item(), m[11].item()])
        print(f"{qb:2d} {w} | {int(m[7].item()):4d} {int(m[6].item()):4d} | {m[10].item():9.0f} {m[11].item():9.0f} | " + " ".join(f"{x:6.0f}" for x in mb) + f" {sum(mb):7.0f} | "
              + " ".join(f"{x:6.0f}" for x in kb) + f" {sum(kb):7.0f} | {m[10].item() - inb:8.0f}")
print("shares (wave 0): main bodies %.3f  masked bodies %.3f  rest of asm %.3f  HIP before/after %.3f" % tuple((tot[:4] / tot[4]).tolist()))
