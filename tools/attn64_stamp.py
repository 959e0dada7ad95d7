"""In-kernel cycle shares of the 64-rows-per-wave forward attention kernel (diagnostic build, never the product):
    python -m musicgeneration_amd._build --variant stamp -DMGX_F2_STAMP       (here, cross-compiles)
    MGX_LIB_PATH=musicgeneration_amd/libmgx_stamp.so python tools/attn64_stamp.py [--B 32]   (GPU box)
Reads the s_memtime sums the stamp build leaves in block A's lse rows and prints, per query-block rank and wave, cycles per
pipelined step split at the step's region boundaries.  (The 64-row dQ kernel this tool also read was removed in round 3
together with the recompute dQ path it belonged to.)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32); ap.add_argument("--L", type=int, default=2048); ap.add_argument("--d", type=int, default=512)
ap.add_argument("--kernel", default="fwd", choices=["fwd"])
a = ap.parse_args()
assert "stamp" in os.environ.get("MGX_LIB_PATH", ""), "load the stamp build: MGX_LIB_PATH=musicgeneration_amd/libmgx_stamp.so"
dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
qkv = (torch.randn(a.B, a.L, 3 * a.d, generator=g) * 0.7).to(torch.bfloat16).to(dev)
E = (torch.randn(a.L, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
dctx = torch.randn(a.B, a.L, a.d, generator=g).to(torch.bfloat16).to(dev)
h = a.d // 64
os.environ["MGX_ATTN_FWD64"] = "1"
for _ in range(3):
    ctx, rec = ops.rel_attn_fwd(qkv, E, None)
torch.cuda.synchronize()
v = rec.float().cpu().view(a.B * h, a.L // 64, 64)[:, :, :14].reshape(-1, 14)
print("qb w  steps |  total  prologue  loop   (per step: region1+barrier  [barrier]  region2)   tail+idle  epilogue   [cycles, mean over (b,h)]")
for qb in range(a.L // 256):
    for w in range(4):
        m = v[(v[:, 9] == qb) & (v[:, 10] == w)]
        n = m[:, 8].mean().item()
        t = m.mean(0)
        per = lambda x: (x / n).item() if n else 0.0
        print(f"{qb:2d} {w}  {int(n):5d} | {t[0]:8.0f} {t[1]:8.0f} {t[2]:8.0f}   ({per(t[3]):7.0f} [{per(t[4]):5.0f}] {per(t[5]):7.0f})   {t[6]:8.0f} {t[7]:8.0f}")
