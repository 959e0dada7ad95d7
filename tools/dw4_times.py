"""Diagnostic (library built with -DMGX_DW4_TIMES): when, inside one launch of linear_dw_ring4_kernel, each workgroup starts, enters and
leaves its main loop and ends (s_memrealtime, 100 MHz), against the launch's duration from HIP events."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(11)
probs = []
for (N, K, has_b) in [(1536, 512, True), (512, 512, True), (256, 512, True), (512, 256, True)]:
    probs.append(((torch.randn(M, N, generator=g) * 0.5).to(dev).bfloat16(), (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16(),
                  torch.zeros(N, K, device=dev), torch.zeros(N, device=dev) if has_b else None))
for _ in range(5):
    ops.linear_dw_grouped(probs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.linear_dw_grouped(probs)
e1.record()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 240))()
lib.mgx_debug_dw4_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.mgx_debug_dw4_times(buf, 8 * 240) == 0
both = np.array(buf, dtype=np.uint64).reshape(240, 8).astype(np.int64)
t, cyc = both[:, :4], both[:, 4:]
lc = cyc[:, 2] - cyc[:, 1]
G = (M // 32 + 11) // 12
print(f"loop cycles (s_memtime): min {lc.min()}  median {int(np.median(lc))}  max {lc.max()}  = {np.median(lc) / G / 32:.1f} cycles per MFMA at {G} stages; "
      f"shader clock in the loop {np.median(lc / ((t[:, 2] - t[:, 1]) / 100.0)) / 1e3:.2f} GHz")
t0 = t[:, 0].min()
us = (t - t0) / 100.0
print(f"grouped dW call (ring4 + fix-up), HIP events: {e0.elapsed_time(e1) * 1e3:.1f} us")
for k, name in enumerate(["start", "loop start", "loop end", "end"]):
    print(f"{name:10s}  min {us[:, k].min():7.1f}  median {np.median(us[:, k]):7.1f}  max {us[:, k].max():7.1f} us after the first workgroup's start")
print(f"loop time   min {(us[:, 2] - us[:, 1]).min():7.1f}  median {np.median(us[:, 2] - us[:, 1]):7.1f}  max {(us[:, 2] - us[:, 1]).max():7.1f} us")
print(f"epilogue    min {(us[:, 3] - us[:, 2]).min():7.1f}  median {np.median(us[:, 3] - us[:, 2]):7.1f}  max {(us[:, 3] - us[:, 2]).max():7.1f} us")
loop = us[:, 2] - us[:, 1]
print("by XCD (workgroup i runs on XCD i % 8): median / max loop time")
for x in range(8):
    sel = loop[x::8]
    print(f"  XCD {x}: {np.median(sel):6.1f} / {sel.max():6.1f} us   units {len(sel)}")
# unit of workgroup i (xcd_remap: XCD x runs units x*30 .. x*30+29): u = (i % 8) * 30 + i // 8; split = u // 20, tile = u % 20
i = np.arange(240)
u = (i % 8) * 30 + i // 8
print("by M-split (20 tiles each):", " ".join(f"{np.median(loop[u // 20 == s]):.0f}" for s in range(12)))
print("by tile:", " ".join(f"{np.median(loop[u % 20 == k]):.0f}" for k in range(20)))
