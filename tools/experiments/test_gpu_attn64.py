"""GPU parity tests of the two alternative forward attention kernels kept under tools/experiments/ (64 rows per wave /
ping-pong; both measured slower than the product kernel, profiles/README.md round 3).  They are NOT part of libmgx.so: build
the experiment variant and point the package at it,

    python -m musicgeneration_amd._build --variant exp --experiments
    MGX_LIB_PATH=musicgeneration_amd/libmgx_exp.so python -m pytest tools/experiments/test_gpu_attn64.py -m gpu -q

Each case is checked twice: against the fp32 oracle on the same bf16-rounded inputs (tolerances of
tests/test_gpu_kernels.py) and against the 32-rows-per-wave kernel of the same library (MGX_ATTN_FWD64=0),
which runs the same arithmetic per tile: the two must agree to bf16 rounding of the outputs."""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.environ.get("MGX_LIB_PATH", "").endswith("libmgx_exp.so"),
                                 reason="needs the experiment build (see the module docstring)")]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


class _env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _inputs(B, L, d, M, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B, L, 3 * d, generator=g) * scale).to(torch.bfloat16)
    E = (torch.randn(M, 64, generator=g) * 0.5).to(torch.bfloat16)
    return qkv, E


@pytest.mark.parametrize("mode", ["1", "2"])            # 1: 64 rows per wave (rel_attn_fwd2.hip), 2: ping-pong (rel_attn_fwd3.hip)
@pytest.mark.parametrize("B,L,d,M,padcase", [(2, 256, 64, 256, 0), (1, 512, 128, 512, 0), (2, 512, 64, 640, 1),
                                              (1, 768, 64, 768, 2), (1, 1024, 128, 1024, 0)])
def test_fwd64_matches_oracle_and_32row_kernel(B, L, d, M, padcase, mode):
    from musicgeneration_amd import ops
    from oracle import ref_cpu as R
    dev = _dev()
    h = d // 64
    qkv, E = _inputs(B, L, d, M, 300 + L + d)
    pad = 7
    tok = torch.randint(0, 7, (B, L), generator=torch.Generator().manual_seed(L), dtype=torch.int32)
    if padcase == 1:
        tok[0, L - 5:] = pad            # trailing pads
        tok[-1, L // 2] = pad           # an isolated padded key in the middle
    if padcase == 2:
        tok[0, L - 300:] = pad          # pads reaching back over a whole 256-row workgroup
    ref_ctx, _, ref_logits = R.attn_core(qkv.float(), E.float(), R.look_ahead_mask(tok, pad), h)
    ref_lse = torch.logsumexp(ref_logits, -1)
    bits = ops.pad_bitmap(tok.to(dev), pad) if padcase else None
    with _env(MGX_ATTN_FWD64=mode):
        ctx, lse = ops.rel_attn_fwd(qkv.to(dev), E.to(dev), bits)
    with _env(MGX_ATTN_FWD64="0"):
        ctx0, lse0 = ops.rel_attn_fwd(qkv.to(dev), E.to(dev), bits)
    torch.cuda.synchronize()
    ctx, lse, ctx0, lse0 = ctx.float().cpu(), lse.cpu(), ctx0.float().cpu(), lse0.cpu()
    assert torch.isfinite(ctx).all() and torch.isfinite(lse).all()
    tol = 2e-2 * ref_ctx.abs().max().item()
    err = (ctx - ref_ctx).abs().max().item()
    assert err <= tol, f"ctx max err {err} > {tol}"
    assert ((ctx - ref_ctx).norm() / ref_ctx.norm()).item() < 1e-2
    assert (lse - ref_lse).abs().max().item() < 2e-3 * max(1.0, ref_lse.abs().max().item())
    # same arithmetic per tile as the 32-row kernel: equal up to one bf16 rounding step of ctx
    assert (ctx - ctx0).abs().max().item() <= 1e-2 * ref_ctx.abs().max().item()
    assert (lse - lse0).abs().max().item() <= 1e-4 * max(1.0, ref_lse.abs().max().item())


@pytest.mark.parametrize("mode", ["1", "2"])
def test_fwd64_softmax_rescale_branch_in_pipelined_loop(mode):
    """Force the lazy-softmax redo INSIDE the pipelined main loop: a late key (third key tile) dominates one query
    row of the second workgroup, so the wave redoes that tile against the true maximum and rescales O once."""
    from musicgeneration_amd import ops
    from oracle import ref_cpu as R
    dev = _dev()
    B, L, d = 1, 512, 64
    qkv, E = _inputs(B, L, d, L, 5, scale=0.3)
    qkv[0, 400, :64] = 4.0      # query row 400: workgroup 1, wave 2 (rows 384..447), block A
    qkv[0, 70, 64:128] = 4.0    # key 70 (key tile 2 < every diagonal of that workgroup): logit = 64*16/8 = 128
    qkv[0, 437, :64] = -4.0     # block B of the same wave: the dominant key is strongly NEGATIVE there (no redo for B)
    tok = torch.zeros(B, L, dtype=torch.int32)
    ref_ctx, _, _ = R.attn_core(qkv.float(), E.float(), R.look_ahead_mask(tok, 9), 1)
    with _env(MGX_ATTN_FWD64=mode):
        ctx, _ = ops.rel_attn_fwd(qkv.to(dev), E.to(dev), None)
    ctx = ctx.float().cpu()
    assert torch.isfinite(ctx).all()
    assert (ctx - ref_ctx).abs().max().item() <= 2e-2 * ref_ctx.abs().max().item()
