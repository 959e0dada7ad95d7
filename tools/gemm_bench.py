"""Micro-benchmark of the GEMM kernels at the cfg2 shapes."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops
dev = torch.device("cuda")
M = int(os.environ.get("GEMM_M", 32768))
shapes = [("qkv", 1536, 512), ("fc", 512, 512), ("ffn_pre", 256, 512), ("ffn_suf", 512, 256), ("vocab", 384, 512)]
g = torch.Generator().manual_seed(0)
def timed(fn, reps=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = {"fwd": 0, "dx": 0, "dw": 0, "dx_step": 0}     # dx_step: the dX calls as the training step makes them (mask / residual addend)
for name, N, K in shapes:
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    dy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    gw = torch.zeros(N, K, device=dev); gb = torch.zeros(N, device=dev)
    fl = 2.0 * M * N * K
    t1 = timed(lambda: ops.linear_fwd(x, w, b, 0)); t2 = timed(lambda: ops.linear_dx(dy, w, None)); t3 = timed(lambda: ops.linear_dw(dy, x, gw, None))
    add = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    t2s = {"qkv": lambda: ops.linear_dx(dy, w, None, add), "ffn_pre": lambda: ops.linear_dx(dy, w, None, add),
           "ffn_suf": lambda: ops.linear_dx(dy, w, add, None)}.get(name)
    t2s = timed(t2s) if t2s else t2
    t4 = timed(lambda: torch.mm(x, w.t()))
    t5 = timed(lambda: torch.mm(dy, w)); t6 = timed(lambda: torch.mm(dy.t(), x))
    byt = 2.0 * (M * K + M * N)
    mult = 6 if name != "vocab" else 1
    tot["fwd"] += t1 * mult; tot["dx"] += t2 * mult; tot["dw"] += t3 * mult; tot["dx_step"] += t2s * mult
    print(f"{name:8s} N={N:5d} K={K:4d}  fwd {t1*1e3:6.1f} us {fl/t1/1e9:6.0f} TF/s | dx {t2*1e3:6.1f} us {fl/t2/1e9:6.0f} (as in the step {t2s*1e3:6.1f}) | dw {t3*1e3:6.1f} us {fl/t3/1e9:6.0f} | lib fwd {t4*1e3:6.1f} dx {t5*1e3:6.1f} dw {t6*1e3:6.1f} | floor hbm {byt/8e12*1e6:5.1f} mfma {fl/2.5e15*1e6:5.1f} us")
# the four weight gradients of one encoder block in one launch (what the training step runs)
probs = []
for name, N, K in shapes[:4]:
    probs.append((torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev), torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev),
                  torch.zeros(N, K, device=dev), torch.zeros(N, device=dev) if name in ("qkv", "ffn_pre") else None))
tg = timed(lambda: ops.linear_dw_grouped(probs))
flg = sum(2.0 * M * N * K for _, N, K in shapes[:4])
print(f"grouped dW of a block: {tg*1e3:6.1f} us {flg/tg/1e9:6.0f} TF/s  (x6 per step = {tg*6:.3f} ms)")
print("per-step totals (ms):", {k: round(v, 3) for k, v in tot.items()})
