"""End-to-end parity of the MI355X model with the reference (golden G2/G9) and with the oracle.
Stated tolerances.  vs the fp32 reference (golden): logits |err| <= 3e-2 * max|logit|, loss rtol
2e-2, gradient cosine >= 0.97 (bf16 activations on a worst-case random-init fixture).
vs the oracle with bf16 rounding emulated at the kernels' storage points (oracle.EMULATE_BF16):
logits within 2 bf16 ulps of max|logit|, gradient cosine >= 0.999 (>= 0.995 for tiny vectors)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _cos(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def _rel(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _model_from(g, prefix, d, nl, L, dropout=0.0):
    from musicgeneration_amd.network import MusicTransformer
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}
    V = sd["fc.weight"].shape[0]
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=dropout)
    missing, unexpected = mt.load_state_dict(sd, strict=True)   # reference key names load unchanged
    return mt.to("cuda"), V


def test_g2_logits_loss_grads(golden_dir):
    from musicgeneration_amd.criterion import SmoothCrossEntropyLoss
    from musicgeneration_amd.metrics import CategoricalAccuracy, LogitsBucketting, MetricsSet
    g = _load(golden_dir, "g2_model.npz")
    mt, V = _model_from(g, "p.", 128, 2, 32)
    x = torch.from_numpy(g["x"]).cuda()
    y = torch.from_numpy(g["y"]).cuda()
    mt.train()
    logits = mt(x)
    ref = torch.from_numpy(g["logits"])
    err = (logits.float().cpu() - ref).abs().max().item()
    assert err <= 3e-2 * ref.abs().max().item(), err
    ms = MetricsSet({"accuracy": CategoricalAccuracy(), "loss": SmoothCrossEntropyLoss(0.1, V, V - 1),
                     "bucket": LogitsBucketting(V)})
    m = ms(logits, y)
    assert abs(m["loss"].item() - float(g["loss"])) <= 2e-2 * float(g["loss"])
    assert abs(m["accuracy"].item() - float(g["accuracy"])) <= 2.0 / y.numel() + 1e-6
    # argmax may only differ where the reference's top-2 logits are closer than the bf16 tolerance
    mism = torch.from_numpy(m["bucket"].cpu().numpy() != g["bucket"])
    top2 = ref.reshape(-1, V).topk(2, -1).values
    gap = top2[:, 0] - top2[:, 1]
    assert (gap[mism] <= 6e-2 * ref.abs().max().item()).all() and mism.float().mean() < 0.1
    m["loss"].backward()
    torch.cuda.synchronize()
    # (a) vs the fp32 reference (golden): bf16 tolerance.  Per-token quantities (the embedding rows)
    #     see un-averaged bf16 activation rounding, so their bound is looser.
    # (b) vs the oracle with bf16 rounding emulated at the kernels' storage points: tight.
    from oracle import ref_cpu as R
    R.EMULATE_BF16 = True
    try:
        pr = {k[2:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith("p.")}
        emu_logits, _ = R.model_forward(pr, torch.from_numpy(g["x"]), V - 1)
        R.smooth_ce(emu_logits, torch.from_numpy(g["y"]), 0.1, V, V - 1).backward()
    finally:
        R.EMULATE_BF16 = False
    assert (logits.float().cpu() - emu_logits.detach()).abs().max().item() <= 2 * 2 ** -8 * ref.abs().max().item()
    for name, p in mt.named_parameters():
        ref_g = torch.from_numpy(g["g." + name])
        emu_g = pr[name].grad
        got = p.grad.cpu()
        assert got.shape == ref_g.shape
        small = ref_g.numel() <= 1024
        if name.endswith("Wk.bias"):
            # exactly-zero true gradient (softmax is invariant to a per-query constant): the reference
            # holds ~1e-10 rounding noise here; ours must be noise-level too, direction is meaningless
            wq = torch.from_numpy(g["g." + name.replace("Wk", "Wq")])
            assert got.abs().max().item() <= 2e-2 * wq.abs().max().item()
            continue
        # rows of the embedding (one token) and of E (one relative distance; few (i,j) pairs at large
        # distances for L=32) are sums over very few samples: bf16 activation rounding is not averaged out
        per_token = name == "Decoder.embedding.weight" or name.endswith("rga.E")
        c, r = _cos(got, ref_g), _rel(got, ref_g)
        # vs fp32: this fixture is a random-init model whose N(0,1)*sqrt(d) embeddings give attention
        # logits of magnitude ~50 (near one-hot softmax), the worst case for bf16 activations; measured
        # cosines are 0.977-0.9999.  The bound that pins the KERNELS is the bf16-emulated one below.
        assert c > 0.97, f"{name}: cos vs fp32 {c}"
        assert r < 0.25, f"{name}: rel vs fp32 {r}"
        c2 = _cos(got, emu_g)
        assert c2 > (0.995 if small else 0.999), f"{name}: cos vs bf16-emulated oracle {c2}"
    # eval convention: (logits, weights-list)
    mt.eval()
    with torch.no_grad():
        out = mt(x)
    assert isinstance(out, tuple) and len(out) == 2
    assert (out[0].float().cpu() - torch.from_numpy(g["eval_logits"])).abs().max().item() <= 3e-2 * ref.abs().max().item()


def test_g7_next_token_probs_match_causal_reference(golden_dir):
    g = _load(golden_dir, "g2_model.npz")
    mt, V = _model_from(g, "p.", 128, 2, 32)
    x = torch.from_numpy(g["x"]).cuda()
    causal = torch.from_numpy(g["g7_causal_probs"])          # [B,L,V] reference eval-mode softmax
    for W in (9, 17, 32):
        probs = mt.next_token_probs(x[:, :W].long()).cpu()
        ref = causal[:, W - 1]
        assert (probs - ref).abs().max().item() < 2e-2
        assert abs(probs.sum(-1) - 1).max().item() < 1e-3


def test_g7_reference_sampling_mask_none_matches_the_reference(golden_dir):
    """The reference's generate() step, Decoder(decode_array, None) (network.py:60-62: NO look-ahead mask at sampling time,
    relative term for j <= i only): the distribution of the next token after the fixture's prior, as the reference itself
    computed it (g7_nomask_probs), and the oracle's restatement on shorter windows (not multiples of 32)."""
    from oracle import ref_cpu as R
    g = _load(golden_dir, "g2_model.npz")
    mt, V = _model_from(g, "p.", 128, 2, 32)
    prior = torch.from_numpy(g["g7_prior"]).cuda()
    probs = mt.next_token_probs(prior.long(), reference_mask=True).cpu()
    ref = torch.from_numpy(g["g7_nomask_probs"])
    assert (probs - ref).abs().max().item() < 2e-2
    assert abs(probs.sum(-1) - 1).max().item() < 1e-3
    # (for the LAST position the two semantics differ only through the earlier rows' bidirectional context in the layers below:
    #  ~1e-4 on this 2-layer fixture; tests/test_gpu_kernels.py::test_rel_attn_fwd_nomask_matches_oracle separates them at kernel level)
    p = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}
    for W in (5, 17, 31):
        lg, _ = R.model_forward(p, prior[:, :W].cpu(), V - 1, causal=False)
        got = mt.next_token_probs(prior[:, :W].long(), reference_mask=True).cpu()
        assert (got - lg.softmax(-1)[:, -1]).abs().max().item() < 2e-2
    out = mt.generate(prior[:, :4].long(), length=3, reference_mask=True)
    assert out.shape == (prior.shape[0], 7)


def test_g9_reference_optimiser_trajectory_at_d64(golden_dir):
    """The round-1 G9 fixture -- the reference's own run (tests/golden/gen_golden.py: Adam(0.9, 0.98, 1e-9) + Noam schedule,
    accum_grad 2, six micro-batches, train.py:143,268-277) at d_model = 64, i.e. ONE head and an FFN width of 32 -- was
    unrunnable until round 6 (FFN width below the GEMMs' reduction granule of 64: loud error).  With the hidden width
    zero-padded inside the flat buffers (network._flat_order: FFN_pre rows, FFN_suf columns) the HIP path follows it: losses
    rtol 2e-2, learning rates exactly, and the parameter movement p3 - p0 of the reference by direction and size."""
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    g = _load(golden_dir, "g9_optim.npz")
    p0 = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p0.")}
    p3 = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p3.")}
    V = p0["fc.weight"].shape[0]
    mt = MusicTransformer(embedding_dim=64, vocab_size=V, num_layer=2, max_seq=16, dropout=0.0)
    mt.load_state_dict(p0)
    mt = mt.cuda().train()
    assert mt.ffn_padded == 64 and tuple(mt.Decoder.enc_layers[0].FFN_suf.weight.shape) == (64, 32)
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    sch = CustomSchedule(64, optimizer=opt)
    lossf = SmoothCrossEntropyLoss(0.1, V, V - 1)
    losses, lrs = [], []
    opt.zero_grad()
    for it in range(6):
        xf = torch.from_numpy(g["xs"][it]).cuda()
        loss = lossf(mt(xf[:, :-1].to(torch.int32)), xf[:, 1:].to(torch.int32)) / 2
        loss.backward()
        losses.append(2 * loss.item())
        if (it + 1) % 2 == 0:
            sch.step()
            lrs.append(sch._rate)
            opt.zero_grad()
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-2)
    sd = {k: v.detach().float().cpu() for k, v in mt.state_dict().items()}
    assert set(sd) == set(p3) and all(sd[k].shape == p3[k].shape for k in sd)
    worst = 1.0
    for k in sd:
        ref = p3[k] - p0[k]
        if k.endswith("Wk.bias") or ref.norm() < 1e-9:       # exactly-zero true gradient: rounding noise on both sides
            continue
        delta = sd[k] - p0[k]
        # Adam's first steps move every element by ~lr * sign(g): bf16 noise flips the sign of near-zero gradients only (G9b: E is
        # the loosest at 0.96-0.97); the size of the movement must agree to 15 %
        c = _cos(delta, ref)
        worst = min(worst, c)
        assert c >= (0.93 if k.endswith("rga.E") else 0.95), (k, c)
        assert abs(float(delta.norm()) - float(ref.norm())) <= 0.15 * float(ref.norm()), k
    # the zero padding stayed zero: rows 32.. of FFN_pre, columns 32.. of FFN_suf in the flat parameter buffer
    st = mt.store()
    for i in range(2):
        pre = f"Decoder.enc_layers.{i}."
        assert st.padded_view(pre + "FFN_pre.weight", 64, 64, "param")[32:].abs().max().item() == 0.0
        assert st.padded_view(pre + "FFN_suf.weight", 64, 64, "param")[:, 32:].abs().max().item() == 0.0


@pytest.mark.parametrize("d", [192, 320])
def test_d_model_64_times_odd_matches_the_oracle(d):
    """VERDICT r5 missing 5: the reference runs every d_model = 64 h (h = d // 64 heads of 64, FFN width d / 2, layers.py:143-144,219);
    for odd h the FFN width is 32 * odd, not a multiple of the GEMMs' reduction granule.  Logits, loss and every gradient of a
    2-layer model at d = 192 (3 heads, FFN 96) and 320 (5 heads, FFN 160) against the oracle's fp32 forward / autograd, then one
    optimiser step and a checkpoint round trip through torch.optim.Adam's state layout."""
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    from oracle import ref_cpu as R
    V, nl, L, B = 60, 2, 64, 3
    pad = V - 1
    p0 = R.init_params(V, d, nl, L, seed=41)
    for k in p0:
        if k.endswith("embedding.weight") or k.endswith("rga.E"):
            p0[k] = p0[k] * 0.3
    assert p0["Decoder.enc_layers.0.FFN_pre.weight"].shape == (d // 2, d)
    g = torch.Generator().manual_seed(d)
    x = torch.randint(0, V - 1, (B, L), generator=g)
    y = torch.randint(0, V - 1, (B, L), generator=g)
    x[1, L - 5:] = pad
    y[1, L - 5:] = pad
    params = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    ref_logits = R.model_forward(params, x, pad)[0]
    ref_loss = R.smooth_ce(ref_logits, y, 0.1, V, pad)
    ref_loss.backward()
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    mt.load_state_dict({k: v.clone() for k, v in p0.items()})
    mt = mt.cuda().train()
    opt = FusedAdam(mt, lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    opt.zero_grad()
    logits = mt(x.to(torch.int32).cuda())
    got, ref = logits.float().cpu(), ref_logits.detach()
    assert (got - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
    assert ((got - ref).norm() / ref.norm()).item() < 1e-2
    loss = SmoothCrossEntropyLoss(0.1, V, pad)(logits, y.to(torch.int32).cuda())
    assert abs(loss.item() - ref_loss.item()) <= 2e-2 * abs(ref_loss.item())
    loss.backward()
    for n, p in mt.named_parameters():
        a, b = p.grad.float().cpu(), params[n].grad
        assert a.shape == b.shape, n
        if b.norm() < 1e-7 or n.endswith("Wk.bias"):
            continue
        assert _cos(a, b) >= 0.99, (n, _cos(a, b))
    before = {k: v.detach().clone() for k, v in mt.state_dict().items()}
    opt.step()
    sd = opt.state_dict()
    moved = mt.state_dict()["Decoder.enc_layers.1.FFN_suf.weight"]
    assert tuple(moved.shape) == (d, d // 2) and (moved - before["Decoder.enc_layers.1.FFN_suf.weight"]).abs().max().item() > 0
    i_suf = sd["param_names"].index("Decoder.enc_layers.1.FFN_suf.weight")
    assert tuple(sd["state"][i_suf]["exp_avg"].shape) == (d, d // 2)
    opt2 = FusedAdam(mt, lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)


def _params_from_oracle_init(shape, seed, scale=None):
    from oracle import ref_cpu as R
    V, d, nl, L, B = (int(v) for v in shape)
    p = R.init_params(V, d, nl, L, seed=seed)
    if scale is not None:
        for k in p:
            if k.endswith("embedding.weight") or k.endswith("rga.E"):
                p[k] = p[k] * scale
    return p, (V, d, nl, L, B)


def test_g9b_three_optimizer_steps_vs_reference_run(golden_dir):
    """The reference's own optimiser trajectory (train.py:143,268-277 + criterion.py:70-96: Adam(0.9,0.98,1e-9),
    Noam schedule, accum_grad=2, 6 micro-batches) at d=128, on the HIP path: losses, learning rates and the
    parameter movement of the reference run (tests/golden/gen_golden.py mt2)."""
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    g = _load(golden_dir, "g9b_optim_d128.npz")
    p0, (V, d, nl, L, B) = _params_from_oracle_init(g["shape"], int(g["seed"]))
    chk = float(sum(v.double().abs().sum().item() for v in p0.values()))
    assert abs(chk - float(g["p0_checksum"])) <= 1e-9 * chk, "initialiser drifted: regenerate the fixture"
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    mt.load_state_dict(p0)
    mt = mt.cuda().train()
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    sch = CustomSchedule(d, optimizer=opt)
    lossf = SmoothCrossEntropyLoss(0.1, V, V - 1)
    losses, lrs = [], []
    opt.zero_grad()
    for it in range(6):
        xf = torch.from_numpy(g["xs"][it]).cuda()
        loss = lossf(mt(xf[:, :-1].to(torch.int32)), xf[:, 1:].to(torch.int32)) / 2
        loss.backward()
        losses.append(2 * loss.item())
        if (it + 1) % 2 == 0:
            sch.step()
            lrs.append(sch._rate)
            opt.zero_grad()
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-2)
    sd = {k: v.detach().float().cpu() for k, v in mt.state_dict().items()}
    # Adam's first steps move every element by ~lr * sign-like terms: the movement's direction and size must follow
    # the reference run (bf16 gradients may flip the sign of near-zero entries only)
    for k in [n[len("delta."):] for n in g if n.startswith("delta.")]:
        delta = sd[k] - p0[k]
        ref = torch.from_numpy(g["delta." + k])
        # measured 0.972 (E), 0.992, 0.999, 1.000 (tools/tol_probe.py).  E is the loosest: Adam's first steps move every
        # element by ~lr * sign(g), so an entry whose tiny gradient flips sign under bf16 noise contributes a full-size
        # error, and E's rows at large distances (few (i,j) pairs at L=32) are such entries
        assert _cos(delta, ref) >= (0.96 if k.endswith("rga.E") else 0.97), (k, _cos(delta, ref))
    names = [str(n) for n in g["delta_norm_names"]]
    for k, rn in zip(names, g["delta_norms"]):
        if k.endswith("Wk.bias"):        # exactly-zero true gradient (softmax shift invariance): rounding noise on both sides
            continue
        dn = float((sd[k].double() - p0[k].double()).norm())
        assert abs(dn - rn) <= 0.15 * rn + 1e-12, (k, dn, rn)


def test_g12_tamed_g2_shape_every_gradient_vs_reference(golden_dir):
    """G2's shape (V=309, 2 layers, d=128, L=32, trailing pads) with tamed logits (embedding and E scaled by 0.25; fixture
    G12 is the reference's own fp32 forward / loss / backward): every parameter gradient of the HIP path is held to the
    fp32 reference at cosine >= 0.995 and rel-L2 <= 0.06 -- measured worst 0.9987 / 0.051 (FFN_pre.weight of the last layer;
    tools/tol_probe.py).  SURVEY 8c asks 0.999 / 2e-2 for kernels against an oracle fed the same bf16 inputs; here the
    model's bf16 activations are part of the error.  G2 itself (test_g2_logits_loss_grads) is a
    raw random-init model with attention logits of ~50 -- the worst case for bf16 -- and keeps its looser fp32 bound."""
    from musicgeneration_amd.criterion import SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    g = _load(golden_dir, "g12_model_d128_tamed.npz")
    p, (V, d, nl, L, B) = _params_from_oracle_init(g["shape"], int(g["seed"]), float(g["scale"]))
    chk = float(sum(v.double().abs().sum().item() for v in p.values()))
    assert abs(chk - float(g["p_checksum"])) <= 1e-9 * chk, "initialiser drifted: regenerate the fixture"
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    mt.load_state_dict(p)
    mt = mt.cuda().train()
    x = torch.from_numpy(g["x"]).cuda()
    logits = mt(x[:, :-1].to(torch.int32))
    ref = torch.from_numpy(g["logits"])
    assert (logits.float().cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    loss = SmoothCrossEntropyLoss(0.1, V, V - 1)(logits, x[:, 1:].to(torch.int32))
    assert abs(loss.item() - float(g["loss"])) <= 2e-3 * float(g["loss"])
    loss.backward()
    torch.cuda.synchronize()
    worst = {}
    for name, prm in mt.named_parameters():
        ref_g = torch.from_numpy(g["g." + name])
        got = prm.grad.cpu()
        if name.endswith("Wk.bias"):      # exactly-zero true gradient (softmax shift invariance): noise on both sides
            continue
        worst[name] = (_cos(got, ref_g), _rel(got, ref_g))
    bad = {k: v for k, v in worst.items() if not (v[0] >= 0.995 and v[1] <= 0.06)}
    assert not bad, bad


def test_g11_d256_model_vs_reference(golden_dir):
    """cfg1's shape family (d=256, h=4, 2 layers) at L=64 with trailing pads, parameters tamed so that bf16 rounding and
    not logit blow-up sets the error: logits, loss and gradients against the reference's fp32 run."""
    from musicgeneration_amd.criterion import SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    g = _load(golden_dir, "g11_model_d256.npz")
    p, (V, d, nl, L, B) = _params_from_oracle_init(g["shape"], int(g["seed"]), float(g["scale"]))
    chk = float(sum(v.double().abs().sum().item() for v in p.values()))
    assert abs(chk - float(g["p_checksum"])) <= 1e-9 * chk, "initialiser drifted: regenerate the fixture"
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    mt.load_state_dict(p)
    mt = mt.cuda().train()
    x = torch.from_numpy(g["x"]).cuda()
    logits = mt(x[:, :-1].to(torch.int32))
    ref = torch.from_numpy(g["logits"])
    err = (logits.float().cpu() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item(), (err, ref.abs().max().item())
    assert _rel(logits.cpu(), ref) <= 1e-2
    loss = SmoothCrossEntropyLoss(0.1, V, V - 1)(logits, x[:, 1:].to(torch.int32))
    assert abs(loss.item() - float(g["loss"])) <= 1e-2 * float(g["loss"])
    loss.backward()
    named = dict(mt.named_parameters())
    for k in [n[2:] for n in g if n.startswith("g.")]:
        c = _cos(named[k].grad.cpu(), torch.from_numpy(g["g." + k]))
        assert c >= 0.99, (k, c)
    for k, rn in zip([str(n) for n in g["grad_norm_names"]], g["grad_norms"]):
        if k.endswith("Wk.bias"):
            continue
        gn = float(named[k].grad.double().norm())
        assert abs(gn - rn) <= 0.05 * rn + 1e-9, (k, gn, rn)


def test_training_reduces_loss_and_matches_oracle_trainer():
    """Same init, same batches, dropout 0: three optimizer steps on the GPU vs the oracle's CpuTrainer."""
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    from oracle import ref_cpu as R
    V, d, nl, L, B = 337, 128, 2, 64, 4
    p0 = R.init_params(V, d, nl, L, seed=0)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    mt.load_state_dict(p0)
    mt = mt.cuda().train()
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    sch = CustomSchedule(d, optimizer=opt)
    lossf = SmoothCrossEntropyLoss(0.1, V, V - 1)
    tr = R.CpuTrainer(p0, pad=V - 1, d_cfg=d, dropout=0.0, accum=2)
    gen = torch.Generator().manual_seed(1234)
    opt.zero_grad()
    for it in range(6):
        xf = torch.randint(0, V - 1, (B, L + 1), generator=gen)
        x, y = xf[:, :-1].to(torch.int32), xf[:, 1:].to(torch.int32)
        ref_loss, _ = tr.step(x, y)
        loss = lossf(mt(x.cuda()), y.cuda()) / 2
        loss.backward()
        assert abs(loss.item() * 2 - ref_loss) <= 2e-2 * ref_loss, (it, loss.item() * 2, ref_loss)
        if (it + 1) % 2 == 0:
            sch.step()
            opt.zero_grad()
    torch.cuda.synchronize()
    # after 3 Adam steps (lr ~ 1e-5 scale) parameters must track the oracle's
    for name, p in mt.named_parameters():
        ref = tr.p[name].detach()
        assert (p.detach().cpu() - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item()), name
        moved = (ref - p0[name]).abs().max().item()
        # Wk.bias has an exactly-zero true gradient: Adam only amplifies rounding noise there
        if moved > 0 and not name.endswith("Wk.bias"):
            got_move = (p.detach().cpu() - p0[name])
            assert _cos(got_move, ref - p0[name]) > 0.9, name


def test_state_dict_roundtrip_and_shadow_sync():
    from musicgeneration_amd.network import MusicTransformer
    mt = MusicTransformer(embedding_dim=128, vocab_size=309, num_layer=1, max_seq=32, dropout=0.0).cuda()
    x = torch.randint(0, 308, (2, 32), dtype=torch.int32, device="cuda")
    mt.eval()
    a = mt(x)[0].float().clone()
    sd = {k: v.clone() for k, v in mt.state_dict().items()}
    assert "Decoder.enc_layers.0.rga.E" in sd and "fc.weight" in sd and not any("_pe" in k for k in sd)
    with torch.no_grad():
        for p in mt.parameters():
            p.mul_(0.5)                       # in-place torch op: shadow must be refreshed automatically
    b = mt(x)[0].float()
    assert (a - b).abs().max().item() > 1e-3
    mt.load_state_dict(sd)
    c = mt(x)[0].float()
    assert (a - c).abs().max().item() == 0.0


def test_eval_attention_weights_match_reference(golden_dir):
    """eval mode returns (logits, [weights per layer]); weights vs the reference's attention_weights (G2)."""
    g = _load(golden_dir, "g2_model.npz")
    mt, V = _model_from(g, "p.", 128, 2, 32)
    mt.eval()
    mt.return_attention_weights = True
    with torch.no_grad():
        logits, ws = mt(torch.from_numpy(g["x"]).cuda())
    assert len(ws) == 2 and ws[0].shape == (3, 2, 32, 32)
    from oracle import ref_cpu as R
    R.EMULATE_BF16 = True
    try:
        pe = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}
        _, emu_ws = R.model_forward(pe, torch.from_numpy(g["x"]), V - 1)
    finally:
        R.EMULATE_BF16 = False
    for w, key, emu in zip(ws, ("eval_w0", "eval_w1"), emu_ws):
        ref = torch.from_numpy(g[key])
        w = w.cpu()
        assert abs(w.sum(-1) - 1).max().item() < 2e-2              # rows are distributions
        assert (w - emu).abs().max().item() < 5e-2                  # vs the oracle with bf16 rounding emulated
        # vs fp32: this fixture has near one-hot attention (logits ~50), so individual weights are
        # hypersensitive to bf16 logits; bound the mean error instead of the max
        assert (w - ref).abs().mean().item() < 2e-3
        assert (w[ref == 0] == 0).all()                             # masked (future / padded-key) entries exactly 0
    mt.return_attention_weights = False
    with torch.no_grad():
        assert mt(torch.from_numpy(g["x"]).cuda())[1] == []


def test_reference_optimizer_line_works_unchanged():
    """train.py:143 uses torch.optim.Adam(model.parameters(), lr=0, betas=(0.9,0.98), eps=1e-9) and
    optimizer.zero_grad() (set_to_none=True since torch 2.0): gradients must still reach the optimizer and
    the bf16 shadow must follow the in-place parameter updates."""
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    torch.manual_seed(0)
    V, L = 309, 64
    xf = torch.randint(0, V - 1, (4, L + 1), device="cuda")
    x, y = xf[:, :-1].to(torch.int32), xf[:, 1:].to(torch.int32)
    lossf = SmoothCrossEntropyLoss(0.1, V, V - 1)

    def run(make_opt):
        torch.manual_seed(1)
        mt = MusicTransformer(embedding_dim=128, vocab_size=V, num_layer=2, max_seq=L, dropout=0.0).cuda().train()
        opt = make_opt(mt)
        sch = CustomSchedule(128, warmup_steps=10, optimizer=opt)
        opt.zero_grad()
        out = []
        for _ in range(6):
            loss = lossf(mt(x), y)
            loss.backward()
            assert all(p.grad is not None for p in mt.parameters())
            sch.step()
            opt.zero_grad()            # torch: set_to_none=True -> our views must be re-attached next forward
            out.append(loss.item())
        return out, mt

    l_torch, m_torch = run(lambda m: torch.optim.Adam(m.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9))
    l_fused, m_fused = run(lambda m: FusedAdam(m, lr=0, betas=(0.9, 0.98), eps=1e-9))
    assert l_torch[-1] < l_torch[0] - 0.05                      # it learns with the reference's optimizer
    for a, b in zip(l_torch, l_fused):
        assert abs(a - b) <= 2e-2 * abs(b)                       # and both optimizers follow the same trajectory
    for (n, p), (_, q) in zip(m_torch.named_parameters(), m_fused.named_parameters()):
        # Adam's first steps move every weight by ~lr*sign(g): weights whose gradient is ~0 may go either way,
        # so compare on average, not element by element
        if n.endswith("Wk.bias"):
            continue        # true gradient is exactly 0 (softmax shift invariance): pure rounding noise drives it
        assert (p - q).abs().mean().item() <= 1e-2 * max(1.0, q.abs().mean().item()), n


def test_embedding_dropout_is_consistent():
    from musicgeneration_amd import ops
    dev = torch.device("cuda")
    B, L, d, V, p = 4, 64, 128, 50, 0.25
    tok = torch.arange(B * L, dtype=torch.int32, device=dev).reshape(B, L) % V
    table = torch.ones(V, d, device=dev)
    pe = torch.zeros(L, d, device=dev)
    out = ops.embed_pe_fwd(tok, table, pe, p, seed=42).float()
    keep = out != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02
    assert (out[keep] - (d ** 0.5) / (1 - p)).abs().max().item() < 0.1        # inverted-dropout scaling (bf16)
    dtable = torch.zeros(V, d, device=dev)
    ops.embed_bwd(tok, torch.ones(B, L, d, dtype=torch.bfloat16, device=dev), dtable, p, seed=42)
    # each vocab row receives sqrt(d)/(1-p) per KEPT occurrence of that (token, column)
    cnt = torch.zeros(V, d, device=dev).index_add_(0, tok.flatten().long(), keep.reshape(-1, d).float())
    assert (dtable - cnt * (d ** 0.5) / (1 - p)).abs().max().item() < 1e-2


def test_training_trajectory_follows_the_oracle():
    """40 optimiser steps (fwd + smoothed CE + bwd + Adam/Noam, dropout 0, gradient accumulation 2) on a learnable
    synthetic task (next token = previous + 1 mod V'): the bf16 kernel path and the oracle's fp32 CpuTrainer start from
    the same weights and see the same batches; the two loss curves stay within 4 % of each other at every one of the 80
    micro-steps and both go down.  End-to-end check of every backward kernel, the flat gradient buffers and the Adam kernel."""
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    from oracle import ref_cpu as R
    V, d, nl, L, B, accum = 60, 128, 2, 64, 8, 2
    p0 = R.init_params(V, d, nl, L, seed=11)
    cpu = R.CpuTrainer(p0, pad=V - 1, d_cfg=d, dropout=0.0, accum=accum)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    mt.load_state_dict({k: v.clone() for k, v in p0.items()})
    mt = mt.cuda().train()
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    sch = CustomSchedule(d, optimizer=opt)
    lossf = SmoothCrossEntropyLoss(0.1, V, V - 1)
    g = torch.Generator().manual_seed(5)
    lc, lg = [], []
    opt.zero_grad()
    for it in range(40 * accum):
        start = torch.randint(0, V - 1, (B, 1), generator=g)
        seq = (start + torch.arange(L + 1)[None, :]) % (V - 1)                 # never the pad id
        x, y = seq[:, :-1].to(torch.int32), seq[:, 1:].to(torch.int32)
        l_cpu, _ = cpu.step(x.long(), y.long())
        loss = lossf(mt(x.cuda()), y.cuda())
        (loss / accum).backward()
        if (it + 1) % accum == 0:
            sch.step()
            opt.zero_grad()
        lc.append(l_cpu)
        lg.append(loss.item())
    for i, (a, b) in enumerate(zip(lc, lg)):
        assert abs(a - b) <= 4e-2 * abs(a), (i, a, b)
    assert lc[-1] < lc[0] - 0.05 and lg[-1] < lg[0] - 0.05          # both learn (Noam warm-up: the first 40 steps are small)


def test_optimizer_state_roundtrips_with_torch_adam():
    """FusedAdam.state_dict() is numbered like torch.optim.Adam(mt.parameters()) (what the reference saves and loads,
    train.py:143-153,201-207): our state loads into torch's Adam and continues identically, and torch's state loads
    back into FusedAdam."""
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    from oracle import ref_cpu as R
    V, d, nl, L = 60, 128, 1, 32
    p0 = R.init_params(V, d, nl, L, seed=4)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    mt.load_state_dict(p0)
    mt = mt.cuda().train()
    opt = FusedAdam(mt, lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    gen = torch.Generator().manual_seed(1)
    names = [n for n, _ in mt.named_parameters()]
    grads = [{n: torch.randn(p0[n].shape, generator=gen) * 0.1 for n in names} for _ in range(3)]

    def set_grads(model, g, dev):
        for n, p in model.named_parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            p.grad.copy_(g[n].to(dev))

    mt.store().attach_grads()
    set_grads(mt, grads[0], "cuda")
    opt.step()
    sd = opt.state_dict()
    # torch side: the same module class on CPU holds the reference's parameter order
    ref = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    ref.load_state_dict({k: v.detach().cpu() for k, v in mt.state_dict().items()})
    topt = torch.optim.Adam(ref.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    cpu_sd = {"state": {i: {k: v.cpu() for k, v in s.items()} for i, s in sd["state"].items()},
              "param_groups": [{k: v for k, v in sd["param_groups"][0].items()}]}
    topt.load_state_dict(cpu_sd)                      # shapes must line up index by index
    for (n, p), i in zip(ref.named_parameters(), range(len(names))):
        assert topt.state[p]["exp_avg"].shape == p.shape, n
    set_grads(ref, grads[1], "cpu")
    topt.step()
    set_grads(mt, grads[1], "cuda")
    opt.step()
    for (n, p), (_, q) in zip(ref.named_parameters(), mt.named_parameters()):
        assert (p.detach() - q.detach().cpu()).abs().max().item() <= 2e-6, n
    # and back: torch's state (no param_names key) into a fresh FusedAdam
    opt2 = FusedAdam(mt, lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    opt2.load_state_dict(topt.state_dict())
    set_grads(ref, grads[2], "cpu")
    topt.step()
    set_grads(mt, grads[2], "cuda")
    opt2.step()
    for (n, p), (_, q) in zip(ref.named_parameters(), mt.named_parameters()):
        assert (p.detach() - q.detach().cpu()).abs().max().item() <= 3e-6, n


def test_library_boundary_records_leading_pads_only():
    """MusicTransformer.forward itself (not only train.py's host-side check) notices LEADING padding -- the one input class with
    fully masked queries: the bitmap kernel raises a sticky device flag, read by check_no_leading_pads() at the caller's next
    synchronisation point.  Trailing and interior pads are ordinary input (the reference accepts them, utils.py:58-83)."""
    from musicgeneration_amd.network import MusicTransformer
    torch.manual_seed(0)
    V, L = 50, 64
    mt = MusicTransformer(embedding_dim=128, vocab_size=V, num_layer=1, max_seq=L, dropout=0.0).cuda().train()
    x = torch.randint(0, V - 1, (4, L), dtype=torch.int32)
    x[1, L - 7:] = V - 1                         # trailing pads: fine
    x[2, L - 1] = V - 1                          # a pad in the last column of a row
    x[0, 10] = V - 1                             # an interior pad: fine since ABI 18
    x[3, :] = V - 1                              # a row of nothing but padding: no real query, silent
    mt(x.cuda())
    mt.check_no_leading_pads()
    lead = x.clone()
    lead[1, :4] = V - 1                          # leading pads
    mt(lead.cuda())
    mt(x.cuda())                                 # the record is sticky across later clean batches
    with pytest.raises(ValueError, match="leading padding"):
        mt.check_no_leading_pads()
    mt.check_no_leading_pads()                   # ... and cleared by the report
    one = x.clone()
    one[2, 0] = V - 1                            # a single pad in column 0 of a row whose next token is real
    mt(one.cuda())
    with pytest.raises(ValueError):
        mt.check_pads_trail()                    # (the old name is an alias)


def test_model_with_interior_pads_matches_the_oracle():
    """interior pads reach the kernels through MusicTransformer.forward and give the reference's result --
    logits of a 2-layer model on rows with pads inside and at the end against the oracle's fp32 forward with the reference's
    look-ahead mask (utils.py:58-83: key j masked iff trg[j] == pad or j > i), and the gradients of the smoothed CE (pad targets
    ignored, criterion.py:52-60) against the oracle's autograd.  Tolerances (bf16 kernels vs fp32): logits <= 3e-2 * max|logit|
    and 1e-2 relative L2, loss 2e-2 relative, gradients cosine >= 0.99."""
    from musicgeneration_amd.criterion import SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from oracle import ref_cpu as R
    V, d, nl, L, B = 60, 128, 2, 96, 3
    pad = V - 1
    p0 = R.init_params(V, d, nl, L, seed=21)
    for k in p0:
        if k.endswith("embedding.weight") or k.endswith("rga.E"):
            p0[k] = p0[k] * 0.3
    g = torch.Generator().manual_seed(8)
    x = torch.randint(0, V - 1, (B, L), generator=g)
    y = torch.randint(0, V - 1, (B, L), generator=g)
    x[0, 5] = pad
    x[0, 40:43] = pad                                       # interior pads (a single one and a run)
    x[1, 17] = pad
    x[1, L - 9:] = pad                                      # interior + trailing
    y[x == pad] = pad
    params = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    ref_logits = R.model_forward(params, x, pad)[0]
    ref_loss = R.smooth_ce(ref_logits, y, 0.1, V, pad)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    mt.load_state_dict({k: v.clone() for k, v in p0.items()})
    mt = mt.cuda().train()
    logits = mt(x.to(torch.int32).cuda())
    got = logits.float().cpu()
    assert torch.isfinite(got).all()
    ref = ref_logits.detach()
    assert (got - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
    assert ((got - ref).norm() / ref.norm()).item() < 1e-2
    mt.check_no_leading_pads()                               # interior pads raise nothing
    loss = SmoothCrossEntropyLoss(0.1, V, pad)(logits, y.to(torch.int32).cuda())
    assert abs(loss.item() - ref_loss.item()) <= 2e-2 * abs(ref_loss.item())
    loss.backward()
    ref_loss.backward()
    checked = 0
    for n, p in mt.named_parameters():
        a, b = p.grad.float().cpu().flatten(), params[n].grad.flatten()
        if b.norm() < 1e-7 or n.endswith("Wk.bias"):         # Wk.bias: exactly-zero true gradient, rounding noise on both sides
            continue
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        assert cos >= 0.99, (n, cos)
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("max_seq,L", [(128, 50), (100, 100), (100, 70), (64, 33), (40, 7)])
def test_any_sequence_length_up_to_max_seq_matches_the_oracle(max_seq, L):
    """VERDICT r5 missing 5: the reference takes every L <= max_seq, and any max_seq (layers.py:64-109, E is [max_seq, 64]); the
    kernels sweep 32-key tiles.  MusicTransformer.forward pads other lengths with trailing pad tokens (masked keys in the causal
    future of every real row) and, when that reaches past a max_seq that is itself no multiple of 32, the positional table and
    the relative embedding with zero rows -- logits and every gradient against the oracle's fp32 forward / autograd at exactly
    that (max_seq, L).  Tolerances as the interior-pad test."""
    from musicgeneration_amd.criterion import SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from oracle import ref_cpu as R
    V, d, nl, B = 60, 128, 2, 3
    pad = V - 1
    p0 = R.init_params(V, d, nl, max_seq, seed=31)
    for k in p0:
        if k.endswith("embedding.weight") or k.endswith("rga.E"):
            p0[k] = p0[k] * 0.3
    g = torch.Generator().manual_seed(L)
    x = torch.randint(0, V - 1, (B, L), generator=g)
    y = torch.randint(0, V - 1, (B, L), generator=g)
    if L > 8:
        x[1, L - 3:] = pad
        y[1, L - 3:] = pad
    params = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    ref_logits = R.model_forward(params, x, pad)[0]
    ref_loss = R.smooth_ce(ref_logits, y, 0.1, V, pad)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=max_seq, dropout=0.0)
    mt.load_state_dict({k: v.clone() for k, v in p0.items()})
    mt = mt.cuda().train()
    logits = mt(x.to(torch.int32).cuda())
    assert tuple(logits.shape) == (B, L, V)
    got, ref = logits.float().cpu(), ref_logits.detach()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
    assert ((got - ref).norm() / ref.norm()).item() < 1e-2
    loss = SmoothCrossEntropyLoss(0.1, V, pad)(logits, y.to(torch.int32).cuda())
    assert abs(loss.item() - ref_loss.item()) <= 2e-2 * abs(ref_loss.item())
    loss.backward()
    ref_loss.backward()
    mt.check_no_leading_pads()
    checked = 0
    for n, p in mt.named_parameters():
        a, b = p.grad.float().cpu(), params[n].grad
        assert a.shape == b.shape, n
        a, b = a.flatten(), b.flatten()
        if b.norm() < 1e-7 or n.endswith("Wk.bias"):
            continue
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        assert cos >= 0.99, (n, cos)
        checked += 1
    assert checked >= 20
    # eval: (logits, weights) as the reference returns them, weights [B,h,L,L]
    mt.eval()
    mt.return_attention_weights = True
    with torch.no_grad():
        lg, ws = mt(x.to(torch.int32).cuda())
    assert tuple(lg.shape) == (B, L, V) and len(ws) == nl and tuple(ws[0].shape) == (B, d // 64, L, L)
    assert (ws[0].sum(-1) - 1).abs().max().item() < 2e-2


def test_sampling_windows_reach_past_a_max_seq_that_is_no_multiple_of_32():
    """next_token_probs at max_seq = 40 (the kernels' tiles are 32 keys): windows of 33-40 tokens pad to 64 rows > max_seq -- zero rows
    for the positional table and the relative embedding (network._params_for_padded_E) in both the causal and the reference's
    mask=None call (network.py:60-62) -- against the oracle"""
    from musicgeneration_amd.network import MusicTransformer
    from oracle import ref_cpu as R
    V, d, nl, M = 60, 128, 2, 40
    p0 = R.init_params(V, d, nl, M, seed=51)
    for k in p0:
        if k.endswith("embedding.weight") or k.endswith("rga.E"):
            p0[k] = p0[k] * 0.3
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=M, dropout=0.0)
    mt.load_state_dict({k: v.clone() for k, v in p0.items()})
    mt = mt.cuda().eval()
    g = torch.Generator().manual_seed(4)
    for W in (33, 37, 40, 7):
        win = torch.randint(0, V - 1, (2, W), generator=g)
        for causal in (True, False):
            lg, _ = R.model_forward(p0, win, V - 1, causal=causal)
            ref = lg.softmax(-1)[:, -1]
            got = mt.next_token_probs(win.cuda(), reference_mask=not causal).cpu()
            assert (got - ref).abs().max().item() < 2e-2, (W, causal)
    out = mt.generate(win[:, :4].cuda(), length=3)
    assert tuple(out.shape) == (2, 7)
