"""Forward GEMM rate vs reduction length (separates per-tile fixed cost from main-loop rate)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops
dev = torch.device("cuda")
M = int(os.environ.get("GEMM_M", 32768)); N = int(os.environ.get("GEMM_N", 1536))
def timed(fn, reps=10):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for K in [int(v) for v in os.environ.get("GEMM_KS", "256,512,1024,2048,4096").split(",")]:
    x = torch.randn(M, K).to(torch.bfloat16).to(dev); w = (torch.randn(N, K) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    b = torch.randn(N).to(dev)
    t = timed(lambda: ops.linear_fwd(x, w, b, 0)); tl = timed(lambda: torch.mm(x, w.t()))
    fl = 2.0 * M * N * K
    print(f"K={K:5d} own {t*1e3:8.1f} us {fl/t/1e9:6.0f} TF/s | lib {tl*1e3:8.1f} us {fl/tl/1e9:6.0f} TF/s")
