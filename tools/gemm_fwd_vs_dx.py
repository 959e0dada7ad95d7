"""dX through the transposed-B ring kernel (as the step runs it) against the same product through the plain ring kernel on a
pre-transposed weight: is a second, transposed bf16 shadow of every weight worth keeping?   GEMM_M=131072 python tools/gemm_fwd_vs_dx.py
Each pair is timed interleaved (dx, nt, dx, nt ...), three rounds."""
import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from musicgeneration_amd import ops
dev = torch.device("cuda"); M = int(os.environ.get("GEMM_M", 131072))
g = torch.Generator().manual_seed(0)
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
tot = [0.0, 0.0]
for name, (N, K) in (("qkv", (1536, 512)), ("fc", (512, 512)), ("ffn_pre", (256, 512)), ("ffn_suf", (512, 256))):
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)       # forward weight [N, K]
    wt = w.t().contiguous()                                                              # [K, N]: dX = dY @ W = linear_fwd(dY, W^T)
    dy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    a = ops.linear_dx(dy, w, None); b = ops.linear_fwd(dy, wt, None, 0)
    assert (a.float() - b.float()).abs().max().item() < 0.1
    fns = (lambda: ops.linear_dx(dy, w, None), lambda: ops.linear_fwd(dy, wt, None, 0))
    for f in fns: f(); f()
    r = [[], []]
    for _ in range(3):
        for i, f in enumerate(fns): r[i].append(t(f))
    d, n = min(r[0]), min(r[1])
    tot[0] += d; tot[1] += n
    print(f"{name:8s} dY [{M} x {N}] . W [{N} x {K}]:  transposed-B kernel {d:7.1f} us   plain kernel on W^T {n:7.1f} us   ({100*(n/d-1):+.1f} %)")
print(f"per block: {tot[0]:.1f} vs {tot[1]:.1f} us  -> {6*(tot[0]-tot[1])/1e3:.3f} ms per 6-layer step")
