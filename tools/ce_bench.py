"""Micro-benchmark of the label-smoothed CE kernels at the cfg2 bench shape (rows = 64 x 2048, V = 337 in rows of 384)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgeneration_amd import ops
dev = torch.device("cuda")
rows, V, ld = 64 * 2048, 337, 384
g = torch.Generator().manual_seed(1)
base = torch.zeros(rows, ld, dtype=torch.bfloat16)
base[:, :V] = (torch.randn(rows, V, generator=g) * 3).to(torch.bfloat16)
base, tgt = base.to(dev), torch.randint(0, V - 1, (rows,), generator=g, dtype=torch.int32).to(dev)
def timed(fn, name, reps=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:10s} {1e3 * e0.elapsed_time(e1) / reps:8.1f} us")
s, a, l = ops.smooth_ce_fwd(base, tgt, V, 0.1, V - 1)
timed(lambda: ops.smooth_ce_fwd(base, tgt, V, 0.1, V - 1), "ce_fwd")
timed(lambda: ops.smooth_ce_bwd(base, tgt, s, l, V, 0.1, V - 1, 1.0), "ce_bwd")
