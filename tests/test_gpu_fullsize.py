"""Parity at BASELINE.json's full sizes (cfg2: L=2048, d=512, 8 heads; cfg4: L=4096, d=768): one sequence
against the oracle (the CPU finishes L=2048 x 8 heads in seconds), and size-independent properties at
the full bench batch: causality (bit-exact), linearity in V, row-normalisation, gradient causality."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _cos(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def test_cfg2_shape_forward_backward_vs_oracle():
    from musicgeneration_amd import ops
    from oracle import ref_cpu as R
    B, L, d, h = 1, 2048, 512, 8
    g = torch.Generator().manual_seed(2048)
    qkv = (torch.randn(B, L, 3 * d, generator=g) * 0.6).to(torch.bfloat16)
    E = (torch.randn(L, 64, generator=g) * 0.3).to(torch.bfloat16)
    dctx = torch.randn(B, L, d, generator=g).to(torch.bfloat16)
    pad = 9
    tok = torch.zeros(B, L, dtype=torch.int32)
    tok[0, L - 100:] = pad                      # trailing pads as a real batch would have
    torch.set_num_threads(8)
    qr, Er = qkv.float().requires_grad_(True), E.float().requires_grad_(True)
    ref, _, logits = R.attn_core(qr, Er, R.look_ahead_mask(tok, pad), h)
    (ref * dctx.float()).sum().backward()
    dev = torch.device("cuda")
    bits = ops.pad_bitmap(tok.to(dev), pad)
    ctx, lse = ops.rel_attn_fwd(qkv.to(dev), E.to(dev), bits)
    dE = torch.zeros(L, 64, device=dev)
    dqkv = ops.rel_attn_bwd(qkv.to(dev), E.to(dev), bits, ctx, dctx.to(dev), lse, dE)
    torch.cuda.synchronize()
    assert (ctx.float().cpu() - ref.detach()).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert (lse.cpu() - torch.logsumexp(logits.detach(), -1)).abs().max().item() < 5e-3
    for name, lo in (("dq", 0), ("dk", d), ("dv", 2 * d)):
        got, want = dqkv.float().cpu()[..., lo:lo + d], qr.grad[..., lo:lo + d]
        assert _cos(got, want) > 0.999 and _rel(got, want) < 2e-2, name
    assert _cos(dE.cpu(), Er.grad) > 0.999 and _rel(dE.cpu(), Er.grad) < 2e-2


# (64, 2048, 512) is bench.py's default per-GPU shape: the first one whose dS workspace (2.18 GB) has byte offsets beyond 2^31
# (4, 4096, 768) is cfg4's single-GPU share, the shape of bench.py's cfg4 block
@pytest.mark.parametrize("B,L,d", [(8, 2048, 512), (64, 2048, 512), (2, 4096, 768), (4, 4096, 768)])
def test_full_batch_properties(B, L, d):
    from musicgeneration_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(7)
    qkv = (torch.randn(B, L, 3 * d, generator=g, device=dev) * 0.6).to(torch.bfloat16)
    E = (torch.randn(L, 64, generator=g, device=dev) * 0.3).to(torch.bfloat16)
    ctx, lse = ops.rel_attn_fwd(qkv, E, None)
    assert torch.isfinite(ctx.float()).all() and torch.isfinite(lse).all()
    # (1) causality, bit-exact: changing every token from position k on leaves rows < k untouched
    k = L // 2 + 37
    q2 = qkv.clone()
    q2[:, k:] = (torch.randn(B, L - k, 3 * d, generator=g, device=dev)).to(torch.bfloat16)
    ctx2, lse2 = ops.rel_attn_fwd(q2, E, None)
    assert torch.equal(ctx[:, :k], ctx2[:, :k]) and torch.equal(lse[:, :, :k], lse2[:, :, :k])
    assert not torch.equal(ctx[:, k:], ctx2[:, k:])
    # (2) the output is linear in V and each row is a convex combination: V = const -> ctx = const
    q3 = qkv.clone()
    q3[..., 2 * d:] = 1.0
    c3, _ = ops.rel_attn_fwd(q3, E, None)
    assert (c3.float() - 1.0).abs().max().item() < 1e-2
    q4 = qkv.clone()
    q4[..., 2 * d:] = qkv[..., 2 * d:] * 2
    c4, _ = ops.rel_attn_fwd(q4, E, None)
    assert _rel(c4.float(), 2 * ctx.float()) < 5e-3
    # (3) gradient causality: a loss on rows < k has zero gradient w.r.t. keys/values/queries at positions >= k,
    #     and dE only touches distances < k
    dctx = torch.zeros(B, L, d, dtype=torch.bfloat16, device=dev)
    dctx[:, :k] = torch.randn(B, k, d, generator=g, device=dev).to(torch.bfloat16)
    dE = torch.zeros(L, 64, device=dev)
    dqkv = ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE)
    torch.cuda.synchronize()
    assert torch.isfinite(dqkv.float()).all() and torch.isfinite(dE).all()
    assert (dqkv[:, k:] == 0).all()
    assert (dE[: L - k] == 0).all() and dE[L - k:].abs().sum().item() > 0      # E row r <-> distance L-1-r < k
    # (4) sum_j dS = 0 per row  =>  a constant added to every K column (q.(k_j + c)) leaves dq's relative part
    #     consistent: sum over the head dim of dk equals the content part of ... (cheap identity) sum_j dv_j = sum_i dO_i
    dv_sum = dqkv[..., 2 * d:].float().sum(1)
    do_sum = dctx.float().sum(1)
    assert _rel(dv_sum, do_sum) < 1e-2


def test_cfg2_model_step_is_finite_and_learns():
    """full cfg2 model, batch 2: loss decreases over a few steps on a fixed batch (end-to-end sanity at size)"""
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    torch.manual_seed(0)
    V, L = 337, 2048
    mt = MusicTransformer(embedding_dim=512, vocab_size=V, num_layer=6, max_seq=L, dropout=0.0).cuda().train()
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    sch = CustomSchedule(512, warmup_steps=20, optimizer=opt)
    lossf = SmoothCrossEntropyLoss(0.1, V, V - 1)
    xf = torch.randint(0, V - 1, (2, L + 1), device="cuda")
    x, y = xf[:, :-1].to(torch.int32), xf[:, 1:].to(torch.int32)
    losses = []
    for _ in range(12):
        loss = lossf(mt(x), y)
        loss.backward()
        sch.step()
        opt.zero_grad()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0] - 0.2, losses


@pytest.mark.parametrize("B", [16, 64])
def test_full_size_dE_and_dQ_two_ways(B):
    """At the bench's per-GPU shape (L=2048, d=512; B=64 is bench.py's default, its dS workspace passes 2^31 bytes): the dE from
    the dS tiles the dK/dV kernel stores equals the full-recompute dE kernel, dQ from those tiles equals the recompute dQ kernel to
    bf16 rounding -- independent implementations of the same sums -- and the LAST batch row's dq/dk/dv equal, bit for bit, a
    batch-of-one call on that row alone (every per-row address, in particular the dS tile offsets, lands where it should)."""
    from musicgeneration_amd import ops
    dev = torch.device("cuda")
    L, d = 2048, 512
    g = torch.Generator().manual_seed(99)
    qkv = (torch.randn(B, L, 3 * d, generator=g) * 0.6).to(torch.bfloat16).to(dev)
    E = (torch.randn(L, 64, generator=g) * 0.3).to(torch.bfloat16).to(dev)
    dctx = torch.randn(B, L, d, generator=g).to(torch.bfloat16).to(dev)
    ctx, lse = ops.rel_attn_fwd(qkv, E, None)
    dE1, dE2 = torch.zeros(L, 64, device=dev), torch.zeros(L, 64, device=dev)
    dq1 = ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE1)                     # pre + dK/dV (stores dS) + dQ and dE from the tiles
    dq2 = ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE2, parts=1 | 32 | 4 | 16)  # dQ and dE by recomputation
    torch.cuda.synchronize()
    assert torch.isfinite(dE1).all() and dE1.abs().max() > 0
    assert _rel(dE1, dE2) < 2e-5
    assert torch.equal(dq1[..., d:], dq2[..., d:])                                  # the same dK/dV kernel on the same inputs
    # two derivations of dS (the dK/dV kernel's and the recompute dQ kernel's own): equal up to the bf16 rounding of dS
    assert _rel(dq1[..., :d].float(), dq2[..., :d].float()) < 4e-3
    # the last row alone
    c1, l1 = ops.rel_attn_fwd(qkv[B - 1:].contiguous(), E, None)
    assert torch.equal(c1[0], ctx[B - 1]) and torch.equal(l1[0], lse[B - 1])
    dEl = torch.zeros(L, 64, device=dev)
    dql = ops.rel_attn_bwd(qkv[B - 1:].contiguous(), E, None, c1, dctx[B - 1:].contiguous(), l1, dEl)
    torch.cuda.synchronize()
    assert torch.equal(dql[0], dq1[B - 1])


def test_full_size_grouped_dW():
    """the grouped dW launch equals four separate launches at the B=16 row count (fp32 accumulation order aside)"""
    from musicgeneration_amd import ops
    dev = torch.device("cuda")
    B, L, d = 16, 2048, 512
    g = torch.Generator().manual_seed(99)
    M = B * L
    shapes = [(3 * d, d), (d, d), (d // 2, d), (d, d // 2)]
    probs, sep = [], []
    for i, (N, K) in enumerate(shapes):
        dy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        gw, gb = torch.zeros(N, K, device=dev), (torch.zeros(N, device=dev) if i % 2 == 0 else None)
        gw2, gb2 = torch.zeros(N, K, device=dev), (torch.zeros(N, device=dev) if i % 2 == 0 else None)
        probs.append((dy, x, gw, gb))
        ops.linear_dw(dy, x, gw2, gb2)
        sep.append((gw2, gb2))
    ops.linear_dw_grouped(probs)
    torch.cuda.synchronize()
    for (dy, x, gw, gb), (gw2, gb2) in zip(probs, sep):
        assert _rel(gw, gw2) < 1e-5
        if gb is not None:
            assert _rel(gb, gb2) < 1e-5


def test_cfg4_shaped_model_step():
    """BASELINE cfg4 as a MODEL (MuMIDI V=486, 12 layers, d=768 = 12 heads, L=4096), batch 1:
    (a) one cfg4-shaped layer + vocabulary projection against the oracle's fp32 forward on the same tokens (trailing pads),
    (b) the full 12-layer model LEARNS at batch 4: ten optimiser steps on a learnable task take the loss down by
        more than 1 nat with finite gradients at every step."""
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    from oracle import ref_cpu as R
    V, d, L = 486, 768, 4096
    pad = V - 1
    p1 = R.init_params(V, d, 1, L, seed=4)
    for k in p1:
        if k.endswith("embedding.weight") or k.endswith("rga.E"):
            p1[k] = p1[k] * 0.2
    g = torch.Generator().manual_seed(44)
    x = torch.randint(0, V - 1, (1, L), generator=g)
    x[0, L - 300:] = pad
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = R.model_forward(p1, x, pad)[0]
    mt1 = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=1, max_seq=L, dropout=0.0)
    mt1.load_state_dict(p1)
    mt1 = mt1.cuda().train()
    with torch.no_grad():
        got = mt1(x.to(torch.int32).cuda()).float().cpu()
    assert torch.isfinite(got).all()
    assert _rel(got, ref) <= 2e-2 and (got - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
    del mt1
    torch.manual_seed(0)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=12, max_seq=L, dropout=0.0).cuda().train()
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    # Noam warm-up 60: the rate climbs by 7.8e-5 per step (7.8e-4 at step 10).  Round 5 ran this test on uniformly random targets
    # with warm-up 20 and saw 6.02 at step 4, 6.42 at step 5: a 12-layer post-LN stack at a rate of 2e-3 five steps after
    # initialisation overshoots (the same schedule at batch 1 happened to pass) -- the schedule's doing, not the kernels'; and random
    # targets leave only ln 485 - 6.35 = -0.17 nat to learn, so that test could not tell a working backward from a broken one.
    sch = CustomSchedule(d, warmup_steps=60, optimizer=opt)
    lossf = SmoothCrossEntropyLoss(0.1, V, pad)
    # per-GPU batch 4 = cfg4's single-GPU share (the shape bench.py's cfg4 block runs: its QKV / output-projection GEMMs take the
    # four-wave ring kernels of the product binary), on the learnable task of test_training_trajectory_follows_the_oracle:
    # next token = previous + 1 (mod V - 1), a fresh random start per row and step
    g = torch.Generator().manual_seed(5)
    losses = []
    for _ in range(10):
        start = torch.randint(0, V - 1, (4, 1), generator=g)
        seq = ((start + torch.arange(L + 1)[None, :]) % (V - 1)).cuda()
        xi, yi = seq[:, :-1].to(torch.int32).contiguous(), seq[:, 1:].to(torch.int32).contiguous()
        loss = lossf(mt(xi), yi)
        loss.backward()
        st = mt.store()
        assert torch.isfinite(st.grad).all(), "non-finite gradient"
        sch.step()
        opt.zero_grad()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses))), losses
    # measured on MI355X (tools/cfg4_learn_probe.py): 6.32, 6.02, 5.59, 4.91, 4.10, 3.24, 2.38, 1.66, 1.21, 0.99 -- the floor of the
    # smoothed loss (eps 0.1, V 486) is 0.95.  Required: a drop of at least 1 nat within the ten steps, and no step going up by more than 0.1
    assert losses[-1] < losses[0] - 1.0, losses
    assert all(b <= a + 0.1 for a, b in zip(losses, losses[1:])), losses


def _det_bench_shape_run(scale):
    """one attention backward + vocabulary-projection dW + embedding gradient + CE statistics at bench.py's per-GPU shape (cfg2,
    batch 64), gradients multiplied by `scale`"""
    from musicgeneration_amd import ops
    dev = torch.device("cuda")
    B, L, d, V = 64, 2048, 512, 337
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B, L, 3 * d, generator=g) * 0.6).to(torch.bfloat16).to(dev)
    E = (torch.randn(L, 64, generator=g) * 0.3).to(torch.bfloat16).to(dev)
    dctx = (torch.randn(B, L, d, generator=g) * scale).to(torch.bfloat16).to(dev)
    tok = torch.randint(0, V - 1, (B, L), generator=g, dtype=torch.int32).to(dev)
    dy = (torch.randn(B * L, 384, generator=g) * scale).to(torch.bfloat16).to(dev)
    x = torch.randn(B * L, d, generator=g).to(torch.bfloat16).to(dev)
    ctx, lse = ops.rel_attn_fwd(qkv, E, None)

    def run():
        dE = torch.zeros(L, 64, device=dev)
        ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE)
        dtab = torch.zeros(V, d, device=dev)
        ops.embed_bwd(tok, dctx, dtab, 0.2, 11)
        gw, gb = torch.zeros(384, d, device=dev), torch.zeros(384, device=dev)
        ops.linear_dw(dy, x, gw, gb)
        torch.cuda.synchronize()
        return dE, dtab, gw, gb
    return run


@pytest.mark.parametrize("scale", [1.0, 1e-6])
def test_deterministic_mode_at_the_bench_shape(scale):
    """deterministic-reduction mode at cfg2 / batch 64 -- two runs bit-equal, and against the default (fp32
    atomics) within 1e-5.  scale = 1e-6 shows what the FIXED 2^-30 quantum costs on tiny gradients: a partial sum of magnitude m
    keeps m * 2^30 integer steps, so the bound there is absolute (quantum x number of partials), stated below."""
    from musicgeneration_amd import ops
    run = _det_bench_shape_run(scale)
    plain = run()
    ops.set_deterministic(True)
    try:
        d1, d2 = run(), run()
    finally:
        ops.set_deterministic(False)
    for a, b, p, name, nparts in zip(d1, d2, plain, ("dE", "dtable", "gW", "gb"), (4096, 4096, 512, 512)):
        assert torch.equal(a, b), name
        assert torch.isfinite(a).all(), name
        # |fixed-point - fp32| <= half a quantum per partial sum (nparts bounds the partials that meet in one element) + the
        # default path's own fp32 summation noise (1e-5 relative)
        bound = 1e-5 * p.abs().max().item() + 0.5 * 2.0 ** -30 * nparts
        assert (a - p).abs().max().item() <= bound, (name, (a - p).abs().max().item(), bound)
        if scale == 1.0:
            assert _rel(a, p) < 1e-5, (name, _rel(a, p))
