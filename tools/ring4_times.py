"""Diagnostic (library built with -DMGX_DW4_TIMES): per workgroup of linear_ring4_kernel, the shader cycles inside the tile statements and
inside the epilogues.   MGX_LIB_PATH=musicgeneration_amd/libmgx_dw4_times.so python tools/ring4_times.py [M N K]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import _lib, ops  # noqa: E402

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (131072, 1536, 512)
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(3)
x = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
w = (torch.randn(N, K, generator=g) * 0.05).to(dev).bfloat16()
b = torch.randn(N, generator=g).to(dev)
lib = _lib.load()
lib.mgx_debug_dw4_times.argtypes = [ctypes.c_void_p, ctypes.c_int]


def report(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * 256))()
    assert lib.mgx_debug_dw4_times(buf, 8 * 256) == 0
    r = np.array(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)
    tiles = r[:, 2]
    print(f"{name}: {e0.elapsed_time(e1) * 1e3:.1f} us; tiles per workgroup {tiles.min():.0f}-{tiles.max():.0f}; per tile: statement "
          f"{np.median(r[:, 0] / tiles):.0f} cycles, epilogue {np.median(r[:, 1] / tiles):.0f} cycles; workgroup lifetime median {np.median(r[:, 4]) / 100:.1f} "
          f"max {r[:, 4].max() / 100:.1f} us; clock {np.median((r[:, 0] + r[:, 1]) / (r[:, 4] / 100)) / 1e3:.2f} GHz")


report(f"fwd  M={M} N={N} K={K}", lambda: ops.linear_fwd(x, w, b, 0))
dy = (torch.randn(M, N, generator=g) * 0.5).to(dev).bfloat16()
add = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
report(f"dx   M={M} N={N} K={K}", lambda: ops.linear_dx(dy, w, None))
report(f"dx+addend", lambda: ops.linear_dx(dy, w, None, add))
