"""In-kernel cycle shares of the 64-key dK/dV kernel with the generated asm main loop (diagnostic build, never the product):
    python -m musicgeneration_amd._build --variant dkv64stamp -DMGX_DKV64_STAMP        (here, cross-compiles)
    MGX_LIB_PATH=musicgeneration_amd/libmgx_dkv64stamp.so python tools/dkv64_stamp.py [--B 64]   (GPU box)
Lane 0 of every wave leaves its s_memtime sums in its first dk row: the five segments of a loop iteration (stamps after the shadows
of MFMAs 10, 16, 28, 36, 44 -- gen_dkv_asm.py STAMP_GAPS) and the kernel's phases (diagonal HIP steps | asm block | epilogue)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64); ap.add_argument("--L", type=int, default=2048); ap.add_argument("--d", type=int, default=512)
a = ap.parse_args()
assert "stamp" in os.environ.get("MGX_LIB_PATH", ""), "load the stamp build: MGX_LIB_PATH=musicgeneration_amd/libmgx_dkv64stamp.so"
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
print(f"stamped dkv64 kernel: {e0.elapsed_time(e1):.3f} ms")
h = a.d // 64
# record: first dk row of every wave's FIRST key tile (64-key waves: key tiles 0, 2, 4, ...), 16 floats
raw = dqkv.view(a.B, a.L // 64, 64, 3 * a.d)[:, :, 0, a.d:2 * a.d].contiguous().view(a.B, a.L // 64, h, 64)
rec = raw.contiguous().view(torch.uint8).view(a.B, a.L // 64, h, 128)[..., :64].contiguous().view(torch.float32).view(-1, 16).cpu()
ok = rec[:, 15] > 0
clk = (rec[ok, 14].sum() / rec[ok, 15].sum()).item()
print(f"s_memtime ticks per s_memrealtime tick (100 MHz): {clk:.2f}  -> {clk * 100:.0f} MHz during this kernel")
print("per key block, cycles:  kb w  nT main-iters |  asm block   epilogue      total | main loop/iter  segs: ->10(barrier)  ->16  ->28  ->36  ->44 | asm block outside the main bodies (prologue, fill, masked bodies)")
tot = torch.zeros(4)
for kb in range(a.L // 128):
    for w in range(2):
        m = rec[(rec[:, 9] == kb) & (rec[:, 10] == w)].mean(0)
        it = m[7].item()
        loop = m[:5].sum().item()
        per = [m[k].item() / max(it, 1) for k in range(5)]
        if w == 0:
            tot += torch.stack([m[12], m[13], m[14], torch.tensor(loop)])
        extra = f" | DMA wait {m[5].item() / max(it, 1):6.0f}  barrier {m[6].item() / max(it, 1):6.0f} per iteration" if m[5].item() + m[6].item() > 0 else ""
        print(f"   {kb:2d} w{w} {int(m[8].item()):4d} {int(it):4d} | {m[12].item():10.0f} {m[13].item():10.0f} {m[14].item():10.0f} | {loop / max(it, 1):8.0f}   "
              + " ".join(f"{p:7.0f}" for p in per) + f" | {m[12].item() - loop:8.0f}" + extra)
print("sum over key blocks (wave 0): asm %.0f (main bodies %.0f)  epilogue %.0f  total %.0f   shares: main bodies %.3f  rest of asm %.3f  epilogue %.3f"
      % (tot[0], tot[3], tot[1], tot[2], tot[3] / tot[2], (tot[0] - tot[3]) / tot[2], tot[1] / tot[2]))
