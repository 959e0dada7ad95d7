"""The four-wave generated-asm weight-gradient ring kernel (linear_dw_ring4_kernel) against the eight-wave HIP kernel on the same
inputs: run once per setting of MGX_DW_RING4 (an experiment-build knob, read once per process), prints a digest of every gradient
and the time per call.   MGX_LIB_PATH=musicgeneration_amd/libmgx_ringab.so MGX_DW_RING4=0|1 python tools/check_dw4.py [M]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(11)
for shapes in ([(1536, 512, True), (512, 512, True), (256, 512, True), (512, 256, True)], [(256, 256, True)], [(768, 256, False), (256, 768, True)]):
    probs = []
    for (N, K, has_b) in shapes:
        dy = (torch.randn(M, N, generator=g) * 0.5).to(dev).bfloat16()
        x = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
        gw = torch.zeros(N, K, device=dev)
        gb = torch.zeros(N, device=dev) if has_b else None
        probs.append((dy, x, gw, gb))
    ops.linear_dw_grouped(probs)
    torch.cuda.synchronize()
    dig = hashlib.sha256()
    worst = 0.0
    for dy, x, gw, gb in probs:
        dig.update(gw.cpu().numpy().tobytes())
        sub = slice(0, min(M, 8192))
        # spot check against fp64 on a row block of the weight (the whole product in fp64 is slow at this M)
        ref = dy[:, :64].double().t() @ x.double()
        worst = max(worst, ((gw[:64].double() - ref).abs().max() / ref.abs().max()).item())
        if gb is not None:
            refb = dy.double().sum(0)
            worst = max(worst, ((gb.double() - refb).abs().max() / refb.abs().max()).item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.linear_dw_grouped(probs)
    e0.record()
    for _ in range(20):
        ops.linear_dw_grouped(probs)
    e1.record()
    torch.cuda.synchronize()
    print(f"M={M} {shapes}: gW digest {dig.hexdigest()[:16]}  worst rel err vs fp64 {worst:.2e}  {e0.elapsed_time(e1) / 20 * 1e3:.1f} us/call", flush=True)
