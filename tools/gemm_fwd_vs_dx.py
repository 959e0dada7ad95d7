import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from musicgeneration_amd import ops
dev = torch.device("cuda"); M = 65536
g = torch.Generator().manual_seed(0)
def t(fn, n=10):
    fn(); fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (N, K) in ((1536, 512), (512, 1536), (512, 512), (256, 512), (512, 256)):
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    dy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    fl = 2.0 * M * N * K
    tf = t(lambda: ops.linear_fwd(x, w, None, 0)); td = t(lambda: ops.linear_dx(dy, w, None))
    print(f"N={N:5d} K={K:5d}  fwd (reduce {K:4d}) {tf:7.1f} us {fl/tf/1e6:6.0f} TF/s | dx (reduce {N:4d}) {td:7.1f} us {fl/td/1e6:6.0f} TF/s")
