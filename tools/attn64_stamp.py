"""In-kernel cycle shares of the 64-rows-per-wave attention kernels (diagnostic build, never the product):
    python -m musicgeneration_amd._build --variant stamp -DMGX_F2_STAMP -DMGX_B2_STAMP       (here, cross-compiles)
    MGX_LIB_PATH=musicgeneration_amd/libmgx_stamp.so python tools/attn64_stamp.py [--kernel fwd|dq] [--B 32]   (GPU box)
Reads the s_memtime sums the stamp build leaves in block A's lse (forward) / delta (dQ) rows and prints, per query-block
rank and wave, cycles per pipelined step split at the step's region boundaries."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32); ap.add_argument("--L", type=int, default=2048); ap.add_argument("--d", type=int, default=512)
ap.add_argument("--kernel", default="dq")
a = ap.parse_args()
assert "stamp" in os.environ.get("MGX_LIB_PATH", ""), "load the stamp build: MGX_LIB_PATH=musicgeneration_amd/libmgx_stamp.so"
dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
qkv = (torch.randn(a.B, a.L, 3 * a.d, generator=g) * 0.7).to(torch.bfloat16).to(dev)
E = (torch.randn(a.L, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
dctx = torch.randn(a.B, a.L, a.d, generator=g).to(torch.bfloat16).to(dev)
h = a.d // 64
if a.kernel == "fwd":
    os.environ["MGX_ATTN_FWD64"] = "1"
    for _ in range(3):
        ctx, rec = ops.rel_attn_fwd(qkv, E, None)
else:
    os.environ["MGX_ATTN_FWD64"] = "0"
    os.environ["MGX_ATTN_DQ64"] = "1"
    ctx, lse = ops.rel_attn_fwd(qkv, E, None)
    lib = ops._lib.load()
    ws = torch.empty(lib.mgx_rel_attn_bwd_workspace(a.B, a.L, a.d), dtype=torch.uint8, device=dev)
    dE = torch.zeros(a.L, 64, device=dev); dqkv = torch.empty_like(qkv)
    for _ in range(3):
        ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 3, dqkv, ws)
    rec = ws[: a.B * h * a.L * 4].view(torch.float32).view(a.B, h, a.L)       # delta sits first in the workspace
torch.cuda.synchronize()
v = rec.float().cpu().view(a.B * h, a.L // 64, 64)[:, :, :14].reshape(-1, 14)
print("qb w  steps |  total  prologue  loop   (per step: region1+barrier  [barrier]  region2)   tail+idle  epilogue   [cycles, mean over (b,h)]")
for qb in range(a.L // 256):
    for w in range(4):
        m = v[(v[:, 9] == qb) & (v[:, 10] == w)]
        n = m[:, 8].mean().item()
        t = m.mean(0)
        per = lambda x: (x / n).item() if n else 0.0
        print(f"{qb:2d} {w}  {int(n):5d} | {t[0]:8.0f} {t[1]:8.0f} {t[2]:8.0f}   ({per(t[3]):7.0f} [{per(t[4]):5.0f}] {per(t[5]):7.0f})   {t[6]:8.0f} {t[7]:8.0f}" + (f"   G1 {per(t[11]):5.0f} G2 {per(t[12]):5.0f} G3 {per(t[13]):5.0f} G4 {per(t[3]-t[4]-t[11]-t[12]-t[13]):5.0f}" if a.kernel == "dq" else ""))
