"""Micro-benchmark of the add + LayerNorm kernels at the training step's shapes (GPU box, repo root):
    python tools/ln_bench.py [--rows 131072] [--d 512] [--p 0.2]
Prints time and bytes/s against the algorithmic bytes of DESIGN.md 2.4 (forward 6d + 8 B per row; backward 8d or 10d with dropout)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=131072); ap.add_argument("--d", type=int, default=512); ap.add_argument("--p", type=float, default=0.2)
ap.add_argument("--reps", type=int, default=20); ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda"); g = torch.Generator().manual_seed(0)
x = torch.randn(a.rows, a.d, generator=g).to(torch.bfloat16).to(dev); res = torch.randn(a.rows, a.d, generator=g).to(torch.bfloat16).to(dev)
dout = torch.randn(a.rows, a.d, generator=g).to(torch.bfloat16).to(dev)
gamma = torch.ones(a.d, device=dev); beta = torch.zeros(a.d, device=dev); dg = torch.zeros(a.d, device=dev); db = torch.zeros(a.d, device=dev)
def timed(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps
out, mean, rstd = ops.add_ln_fwd(x, res, gamma, beta, 1e-6, a.p, 7)
for _ in range(a.rounds):
    tf = timed(lambda: ops.add_ln_fwd(x, res, gamma, beta, 1e-6, a.p, 7))
    tb = timed(lambda: ops.add_ln_bwd(dout, x, res, gamma, mean, rstd, dg, db, a.p, 7))
    bf = a.rows * (6 * a.d + 8); bb = a.rows * ((10 if a.p > 0 else 8) * a.d)
    print(f"add_ln_fwd {tf*1e3:7.1f} us {bf/tf/1e9:6.2f} TB/s | add_ln_bwd {tb*1e3:7.1f} us {bb/tb/1e9:6.2f} TB/s")
