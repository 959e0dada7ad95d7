"""The 256 x 256 ring GEMMs (linear.hip: linear_ring_kernel, eight waves, HIP; linear_ring4_kernel, four waves, generated asm tile
statement) against the 128 x 128 kernels they replace for the big projections:
same inputs through both paths in two child processes (the path is chosen once per process by MGX_GEMM_RING, a knob that
exists in EXPERIMENT builds only: the test builds `libmgx_ringab.so` with `_build.py --experiments` and loads it through
MGX_LIB_PATH; the product library reads no environment variable), outputs
compared BIT FOR BIT (both accumulate the reduction in the same order), and against an fp32 torch reference.
The shapes with 300 and 512 tiles give some of the 256 persistent workgroups two tiles: the ring then runs across a
tile boundary, with the epilogue's stores and the next tile's bias DMA in the in-order queue."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, torch, numpy as np
sys.path.insert(0, sys.argv[1])
from musicgeneration_amd import _lib
from musicgeneration_amd.ops import ptr, stream_ptr, check
lib = _lib.load()
dev = "cuda:0"
out = {}
g = torch.Generator(device="cpu").manual_seed(7)
def rnd(*s): return (torch.randn(*s, generator=g) * 0.5).to(dev).bfloat16()
# forward: (M, N, K, bias, act)
for i, (M, N, K, use_b, act) in enumerate([(512, 256, 128, True, 0), (768, 512, 256, True, 1), (1024, 256, 512, False, 0),
                                           (2048, 1536, 512, True, 0), (256, 256, 192, True, 0),
                                           (76800, 256, 128, True, 1), (131072, 256, 256, True, 0)]):
    A, W = rnd(M, K), rnd(N, K)
    b = (torch.randn(N, generator=g)).to(dev) if use_b else None
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    check(lib.mgx_linear_fwd(ptr(A), ptr(W), ptr(b), ptr(C), M, N, K, act, stream_ptr()), "fwd")
    ref = A.float() @ W.float().t() + (b if use_b else 0)
    if act: ref = ref.relu()
    out[f"fwd{i}"] = C.view(torch.int16).cpu().numpy()
    out[f"fwd{i}_ref"] = ref.cpu().numpy()
# dX: (M, N(reduction), K(out), relu mask, addend)
for i, (M, N, K, use_y, use_add) in enumerate([(512, 128, 256, False, False), (768, 256, 512, True, False),
                                               (1024, 1536, 512, False, True), (512, 512, 256, True, True),
                                               (76800, 128, 256, True, True),
                                               # ring-sized calls with ONE epilogue operand, as the training step makes them (the ring
                                               # kernel prefetches the operand's rows; a call with both goes to the 128 x 128 kernel)
                                               (65536, 512, 256, True, False), (49152, 256, 512, False, True),
                                               (32768, 1536, 512, False, True), (65536, 256, 256, False, False)]):
    dY, W = rnd(M, N), rnd(N, K)
    y = rnd(M, K) if use_y else None
    add = rnd(M, K) if use_add else None
    dX = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    check(lib.mgx_linear_dx(ptr(dY), ptr(W), ptr(y), ptr(add), ptr(dX), M, N, K, stream_ptr()), "dx")
    ref = dY.float() @ W.float()
    if use_y: ref = ref * (y.float() > 0)
    if use_add: ref = ref.bfloat16().float() * 0 + (dY.float() @ W.float()) * ((y.float() > 0) if use_y else 1.0)  # mask first
    out[f"dx{i}"] = dX.view(torch.int16).cpu().numpy()
    out[f"dx{i}_ref"] = ref.cpu().numpy()
    if use_add: out[f"dx{i}_add"] = add.float().cpu().numpy()
torch.cuda.synchronize()
np.savez(sys.argv[2], **out)
'''


_VARIANT = None


def _variant_lib():
    """experiment build of the tracked sources (the product's kernels + the A/B knobs), built once per test session"""
    global _VARIANT
    if _VARIANT is None:
        sys.path.insert(0, ROOT)
        from musicgeneration_amd import _build
        _VARIANT = _build.build(variant="ringab", experiments=True)
    return _VARIANT


def _run(ring, path):
    # MGX_RING4=2: the four-wave asm kernel wherever its shape rules allow (the product takes it only for long streams of stages)
    env = dict(os.environ, MGX_GEMM_RING=str(ring), MGX_RING4="2", MGX_LIB_PATH=_variant_lib())
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(path)


def _bf16(a):
    return (a.astype(np.uint16).astype(np.uint32) << 16).view(np.float32)


def test_ring_matches_tiled_kernels_bitwise(tmp_path):
    old = _run(0, str(tmp_path / "old.npz"))
    new = _run(1, str(tmp_path / "new.npz"))
    for k in old.files:
        if k.endswith("_ref") or k.endswith("_add"):
            continue
        assert np.array_equal(old[k], new[k]), f"{k}: ring kernel differs from the 128x128 kernel"
        got = _bf16(new[k].view(np.uint16))
        ref = new[k + "_ref"]
        if k + "_add" in new.files:
            # the kernel rounds the (masked) product to bf16 and adds the bf16 addend in fp32, then rounds once more
            ref = ref + new[k + "_add"]
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-6)
        assert err < 2e-2, f"{k}: relative error {err:.3e} vs fp32 reference"


@pytest.mark.parametrize("M,shapes", [
    (4096 + 32 * 5, [(768, 256, True), (256, 256, False), (512, 256, True), (256, 512, False)]),   # 12 steps per unit, ragged last split
    (4096, [(256, 256, True)]),                          # one tile, 128 splits of ONE 32-row step (the short-stream paths)
    (12288, [(256, 256, False), (256, 256, True)]),      # 2-3 steps per unit
    # cfg4's block at d = 768: the FFN weights, 384 x 768 and 768 x 384, have a RAGGED last tile row / column (75 % of their tiles)
    (4096, [(2304, 768, True), (768, 768, False), (384, 768, True), (768, 384, True)]),
    # the vocabulary projection alone (448 x 512: second tile row 192 of 256) and with odd multiples of 8
    (8192, [(448, 512, True)]),
    (4096, [(456, 520, True), (264, 248, True)]),
    # a MIXED group: the weights that fill their tiles take the ring kernel, the others (128 x 768: half a tile row) the 128 x 128
    # grouped kernel, from one call
    (4096, [(768, 768, True), (128, 768, True), (768, 384, False)]),
])
def test_dw_ring_against_fp64_reference(M, shapes):
    """weight / bias gradients of a block's four projections through the ring TN kernel + fix-up pass (whole 256 x 256
    tilings, M >= 4096) against an fp64 product of the same bf16 operands; gradients ACCUMULATE into their slots."""
    import torch
    from musicgeneration_amd import ops, _lib
    import ctypes
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(11)
    probs, refs = [], []
    for (N, K, has_b) in shapes:
        dy = (torch.randn(M, N, generator=g) * 0.5).to(dev).bfloat16()
        x = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
        gw = torch.randn(N, K, generator=g).to(dev)
        gb = torch.randn(N, generator=g).to(dev) if has_b else None
        refs.append((gw.double() + dy.double().t() @ x.double(), (gb.double() + dy.double().sum(0)) if has_b else None))
        probs.append((dy, x, gw, gb))
    lib = _lib.load()
    arr = (ops._DwProblem * len(probs))()
    for i, (dy, x, gw, gb) in enumerate(probs):
        arr[i] = ops._DwProblem(ops.ptr(dy), ops.ptr(x), ops.ptr(gw), ops.ptr(gb), gw.shape[0], gw.shape[1])
    need = lib.mgx_linear_dw_grouped_workspace(ctypes.cast(arr, ctypes.c_void_p), len(probs), M)
    assert need > 0, "this group must take the ring kernel (whole 256x256 tiles, M % 32 == 0)"
    ops.linear_dw_grouped(probs)
    torch.cuda.synchronize()
    for (dy, x, gw, gb), (rw, rb) in zip(probs, refs):
        err = (gw.double() - rw).abs().max().item() / rw.abs().max().item()
        assert err < 1e-5, f"gW {tuple(gw.shape)}: relative error {err:.2e}"
        if gb is not None:
            errb = (gb.double() - rb).abs().max().item() / rb.abs().max().item()
            assert errb < 1e-5, f"gb {tuple(gb.shape)}: relative error {errb:.2e}"


CHILD_DW = r'''
import sys, os, torch, numpy as np
sys.path.insert(0, sys.argv[1])
from musicgeneration_amd import ops
dev = "cuda:0"
out = {}
g = torch.Generator(device="cpu").manual_seed(13)
for i, (M, shapes) in enumerate([(4096 + 32 * 5, [(768, 256, True), (256, 256, False), (512, 256, True), (256, 512, True)]),
                                 (4096, [(256, 256, True)]),
                                 (32768, [(1536, 512, True), (512, 512, True), (256, 512, True), (512, 256, True)]),
                                 (8192, [(2304, 768, True), (768, 768, True)])]):
    probs = []
    for (N, K, has_b) in shapes:
        probs.append(((torch.randn(M, N, generator=g) * 0.5).to(dev).bfloat16(), (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16(),
                      torch.randn(N, K, generator=g).to(dev), torch.randn(N, generator=g).to(dev) if has_b else None))
    ops.linear_dw_grouped(probs)
    torch.cuda.synchronize()
    for j, (dy, x, gw, gb) in enumerate(probs):
        out[f"gw{i}_{j}"] = gw.cpu().numpy()
        if gb is not None:
            out[f"gb{i}_{j}"] = gb.cpu().numpy()
            out[f"gbref{i}_{j}"] = dy.double().sum(0).cpu().numpy()
np.savez(sys.argv[2], **out)
'''


def test_dw_ring4_matches_the_eight_wave_kernel_bitwise(tmp_path):
    """the generated-asm four-wave weight-gradient kernel (linear_dw_ring4_kernel, the product path) against the eight-wave HIP ring
    kernel it replaced (MGX_DW_RING4=0, experiment builds only): same images, same stage order, same MFMA operand order -> the weight
    gradients are bit-identical; the bias gradients are summed by different waves in a different order -> close, not identical"""
    res = {}
    for four in (0, 1):
        path = str(tmp_path / f"dw{four}.npz")
        env = dict(os.environ, MGX_DW_RING4=str(four), MGX_LIB_PATH=_variant_lib())
        r = subprocess.run([sys.executable, "-c", CHILD_DW, ROOT, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        res[four] = np.load(path)
    for k in res[0].files:
        if k.startswith("gw"):
            assert np.array_equal(res[0][k], res[1][k]), f"{k}: the four-wave kernel's weight gradient differs"
        elif k.startswith("gb") and not k.startswith("gbref"):
            scale = np.abs(res[0]["gbref" + k[2:]]).max()
            assert np.abs(res[0][k] - res[1][k]).max() < 1e-5 * scale, f"{k}: bias gradient"


# (kind, M, N, K, epilogue, expected kernel family): forward C [M,N] = act(A [M,K] . W [N,K]^T + b); dX [M,K] = dY [M,N] . W [N,K]
# (reduction N) with a ReLU mask / a residual addend.  The shapes are the bench's: cfg2 at batch 32 / 64 (M = 65,536 / 131,072:
# QKV 1536 x 512, output projection 512 x 512, FFN 256 x 512 and 512 x 256) and cfg4 at batch 8 (M = 32,768: QKV 2304 x 768).
_PRODUCT_SHAPES = [
    ("fwd", 65536, 1536, 512, "bias", "RING4"),
    ("fwd", 131072, 512, 512, "bias", "RING4"),
    ("fwd", 32768, 2304, 768, "bias", "RING4"),
    ("fwd", 131072, 256, 512, "bias+relu", "RING8"),       # FFN_pre: 256-column output -> the eight-wave kernel
    ("fwd", 131072, 512, 256, "bias", "RING8"),            # FFN_suf: reduction 256
    ("dx", 131072, 1536, 512, "addend", "RING4"),          # dX of QKV + the residual gradient
    ("dx", 131072, 512, 512, "none", "RING4"),             # dX of the output projection
    ("dx", 65536, 512, 512, "mask", "RING4"),              # (mask-only four-wave variant: not a call site of the model, still a kernel of the binary)
    ("dx", 131072, 512, 256, "mask", "RING8"),             # dX of FFN_suf under FFN_pre's ReLU
    ("dx", 131072, 256, 512, "addend", "RING8"),           # dX of FFN_pre + the residual gradient
    ("dx", 32768, 2304, 768, "addend", "RING4"),
]


@pytest.mark.parametrize("kind,M,N,K,epi,family", _PRODUCT_SHAPES)
def test_product_library_ring_kernels_at_bench_shapes(kind, M, N, K, epi, family):
    """the ring forward / dX kernels of the PRODUCT binary (libmgx.so, no experiment build, no environment
    knob) at shapes that pass its own gates -- the GEMMs bench.py's step really runs -- against an fp32 product of the same
    bf16 operands computed on the GPU.  `mgx_linear_kernel_id` (ABI 18) states which kernel family the call took, so the
    test cannot silently check the 128 x 128 kernels instead.  Tolerance as test_linear_fwd: |err| <= 2^-7 max|ref| + 1e-3
    (one bf16 rounding of the result), and 5e-3 relative L2."""
    import torch
    from musicgeneration_amd import _lib, ops
    assert "MGX_LIB_PATH" not in os.environ, "this test is about the product library"
    lib = _lib.load()
    fam = {"RING8": 2, "RING4": 3}[family]
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(M // 256 + N + K)

    def rnd(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(dev).bfloat16()
    if kind == "fwd":
        assert lib.mgx_linear_kernel_id(0, M, N, K, None) == fam
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, generator=g).to(dev)
        act = 1 if "relu" in epi else 0
        out = ops.linear_fwd(a, w, bias, act)
        ref = a.float() @ w.float().t() + bias
        if act:
            ref = ref.relu()
    else:
        k_id = {"none": 1, "mask": 2, "addend": 3}[epi]
        assert lib.mgx_linear_kernel_id(k_id, M, N, K, None) == fam
        dy, w = rnd(M, N), rnd(N, K, scale=N ** -0.5)
        y = rnd(M, K) if epi == "mask" else None
        add = rnd(M, K) if epi == "addend" else None
        out = ops.linear_dx(dy, w, y, add)
        ref = dy.float() @ w.float()
        if y is not None:
            ref = ref * (y.float() > 0)
        if add is not None:
            # the kernel rounds the product to bf16, adds the bf16 addend in fp32 and rounds once more
            ref = ref + add.float()
    torch.cuda.synchronize()
    got = out.float()
    assert torch.isfinite(got).all()
    tol = 2 ** -7 * ref.abs().max().item() + 1e-3
    if epi == "addend":
        tol *= 2                                            # two roundings
    err = (got - ref).abs().max().item()
    assert err <= tol, f"max err {err} > {tol}"
    assert ((got - ref).norm() / ref.norm()).item() < 5e-3


def test_linear_kernel_id_follows_the_stream_cu_count():
    """the persistent ring kernels take a call when its 256 x 256 tiles fill 3/4 of the CUs of the stream it is issued on: a
    CU-masked stream (mgx_stream_create_cu_mask, ABI 18) with a quarter of the chip sends a quarter-size GEMM down the ring path
    that the whole device would give to the 128 x 128 kernels -- and the result is the same to the bit (same reduction order)"""
    import ctypes
    import torch
    from musicgeneration_amd import _lib, ops
    lib = _lib.load()
    torch.zeros(1, device="cuda")
    M, N, K = 16384, 512, 512                               # 128 tiles: < 192 on the whole device
    assert lib.mgx_linear_kernel_id(0, M, N, K, None) == 1
    side = ops.masked_stream(64)
    if True:
        assert lib.mgx_stream_cus(side.ptr) == 64
        assert lib.mgx_linear_kernel_id(0, M, N, K, side.ptr) == 3
        g = torch.Generator(device="cpu").manual_seed(3)
        a = torch.randn(M, K, generator=g).cuda().bfloat16()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().bfloat16()
        b = torch.randn(N, generator=g).cuda()
        whole = ops.linear_fwd(a, w, b, 0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side.stream):
            masked = ops.linear_fwd(a, w, b, 0)
        side.stream.synchronize()
        assert torch.equal(whole.view(torch.int16), masked.view(torch.int16))
