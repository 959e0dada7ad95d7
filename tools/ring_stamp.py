"""In-kernel cycle shares of the ring GEMM (diagnostic build, never the product):
    python -m musicgeneration_amd._build --variant ringstamp -DMGX_RING_STAMP        (here, cross-compiles)
    MGX_LIB_PATH=musicgeneration_amd/libmgx_ringstamp.so python tools/ring_stamp.py [--M 131072]   (GPU box)
Lane 0 of every wave leaves its s_memtime sums in the first bytes of C (the output is then garbage there): per shape, cycles per
reduction step split into block 1 (8 MFMA + reads) / counted DMA wait / barrier / block 2, and cycles per epilogue."""
import argparse, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=131072)
a = ap.parse_args()
assert "stamp" in os.environ.get("MGX_LIB_PATH", ""), "load the stamp build: MGX_LIB_PATH=musicgeneration_amd/libmgx_ringstamp.so"
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
M = a.M
for name, N, K in [("qkv", 1536, 512), ("fc", 512, 512), ("ffn_pre", 256, 512), ("ffn_suf", 512, 256)]:
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    dy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    add = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    for kind, fn in (("fwd", lambda: ops.linear_fwd(x, w, b, 0)), ("dx", lambda: ops.linear_dx(dy, w, None)),
                     ("dx+add", lambda: ops.linear_dx(dy, w, None, add)), ("dx+mask", lambda: ops.linear_dx(dy, w, add, None))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        rec = out.contiguous().view(torch.uint8).view(-1)[:256 * 8 * 64].clone().view(torch.float32).view(-1, 16).cpu()
        rec = rec[(rec[:, 5] > 0) & (rec[:, 8] > 0)]
        G = rec[:, 5].mean().item(); T = rec[:, 6].mean().item()
        mhz = (rec[:, 7].sum() / rec[:, 8].sum()).item() * 100
        m = rec.mean(0)
        steps = [m[i].item() / G for i in range(4)]
        print(f"{name:8s} {kind:7s} {e0.elapsed_time(e1)*1e3:7.1f} us  clock {mhz:5.0f} MHz  steps/WG {G:5.0f} tiles/WG {T:4.1f} | per step: block1 {steps[0]:6.0f} "
              f"dma-wait {steps[1]:5.0f} barrier {steps[2]:5.0f} block2 {steps[3]:6.0f} = {sum(steps):6.0f} (MFMA floor 1024 per SIMD, 512 per wave) | "
              f"epilogue {m[4].item() / max(T, 1):7.0f} per tile | total {m[7].item():9.0f} cycles")
