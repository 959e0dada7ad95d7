"""In-kernel cycle shares of a main-loop step of the dK/dV attention kernel (diagnostic build, never the product):
    python -m musicgeneration_amd._build --variant dkvstamp -DMGX_DKV_STAMP        (here, cross-compiles)
    MGX_LIB_PATH=musicgeneration_amd/libmgx_dkvstamp.so python tools/dkv_stamp.py [--B 32]   (GPU box)
Reads the s_memtime sums lane 0 of every wave leaves in its first dk row and prints, per key-block rank, cycles per step split at
the stamps of rel_attn_bwd.hip (DKV_STAMP)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32); ap.add_argument("--L", type=int, default=2048); ap.add_argument("--d", type=int, default=512)
a = ap.parse_args()
assert "stamp" in os.environ.get("MGX_LIB_PATH", ""), "load the stamp build: MGX_LIB_PATH=musicgeneration_amd/libmgx_dkvstamp.so"
dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
qkv = (torch.randn(a.B, a.L, 3 * a.d, generator=g) * 0.7).to(torch.bfloat16).to(dev)
E = (torch.randn(a.L, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
dctx = torch.randn(a.B, a.L, a.d, generator=g).to(torch.bfloat16).to(dev)
dE = torch.zeros(a.L, 64, device=dev)
ctx, lse = ops.rel_attn_fwd(qkv, E, None)
dqkv = torch.zeros_like(qkv)
ws = torch.empty(ops._lib.load().mgx_rel_attn_bwd_workspace(a.B, a.L, a.d), dtype=torch.uint8, device=dev)
for _ in range(3):
    ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 1 | 4, dqkv, ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 4, dqkv, ws); e1.record(); torch.cuda.synchronize()
print(f"stamped dkv kernel: {e0.elapsed_time(e1):.3f} ms")
h = a.d // 64
# record: first dk row of every wave's key tile = rows j0 = 32 * tile; columns d + hd*64 .. as raw bytes
raw = dqkv.view(a.B, a.L // 32, 32, 3 * a.d)[:, :, 0, a.d:2 * a.d].contiguous().view(a.B, a.L // 32, h, 64)
rec = raw.contiguous().view(torch.uint8).view(a.B, a.L // 32, h, 128)[..., :56].contiguous().view(torch.float32).view(-1, 14).cpu()
ok = rec[:, 10] > 0
print(f"s_memtime ticks per s_memrealtime tick (100 MHz): {(rec[ok, 9].sum() / rec[ok, 10].sum()).item():.2f}  -> s_memtime counts at "
      f"{(rec[ok, 9].sum() / rec[ok, 10].sum()).item() * 100:.0f} MHz during this kernel")
print("per workgroup (wave 0), cycles: kb  total  prologue  4 diagonal steps  main loop  rest (odd step + epilogue)")
tot = torch.zeros(5)
for kb in range(a.L // 128):
    m = rec[(rec[:, 7] == kb) & (rec[:, 8] == 0)].mean(0)
    row = torch.stack([m[9], m[11], m[12], m[13], m[9] - m[11] - m[12] - m[13]])
    tot += row
    print(f"   {kb:2d} " + " ".join(f"{v.item():10.0f}" for v in row))
print("  sum " + " ".join(f"{v.item():10.0f}" for v in tot) + "   shares " + " ".join(f"{(v / tot[0]).item():.3f}" for v in tot[1:]))
names = ["QE+merge", "bpermute", "S,dP,exp,dS", "dV,dK,stores", "publish+bar", "prefetch"]
print("kb  w  steps | cycles per step: " + "  ".join(f"{n:>12s}" for n in names) + "        sum")
for kb in range(a.L // 128):
    for w in range(4):
        m = rec[(rec[:, 7] == kb) & (rec[:, 8] == w)]
        if not len(m) or m[:, 6].mean().item() == 0: continue
        n = m[:, 6].mean().item(); t = m.mean(0)
        print(f"{kb:2d}  {w}  {int(n):5d} |                  " + "  ".join(f"{(t[i] / n).item():12.0f}" for i in range(6)) + f"   {(t[:6].sum() / n).item():8.0f}")
    if kb >= 3 and kb % 4: continue
