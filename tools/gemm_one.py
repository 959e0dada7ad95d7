"""One GEMM shape, own kernels only (for PMC passes): GEMM_SHAPE=N,K GEMM_M=32768 python tools/gemm_one.py"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops
dev = torch.device("cuda")
M = int(os.environ.get("GEMM_M", 32768))
N, K = (int(v) for v in os.environ.get("GEMM_SHAPE", "1536,512").split(","))
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
b = torch.randn(N, generator=g).to(dev)
dy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
gw = torch.zeros(N, K, device=dev)
for _ in range(int(os.environ.get("GEMM_REPS", 3))):
    ops.linear_fwd(x, w, b, 0); ops.linear_dx(dy, w, None); ops.linear_dw(dy, x, gw, None)
torch.cuda.synchronize()
