"""ORACLE -- test infrastructure only.  NOT part of the product path.

CPU fp32 restatement (eager PyTorch, materialised L x L attention) of the reference's
autoregressive event-sequence hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; ``musicgeneration_amd`` never does.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py`` against golden
vectors captured by importing the reference itself in the build container
(``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``).

Each function cites the reference file:line it restates (paths relative to
``/root/reference/mg/model/MusicTransformer`` unless noted).  The arithmetic is re-derived
(index formulas instead of the reference's pad/reshape skew), not copied.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# bf16 emulation: when True, every tensor the MI355X path stores in bf16 (GEMM weights, E, the
# embedding/LN/GEMM/attention outputs and the attention probabilities) is rounded to bf16 at the
# same point, with fp32 arithmetic in between -- i.e. the oracle computes what the kernels compute
# up to accumulation order.  The fp32 reference semantics are EMULATE_BF16 = False (default).
EMULATE_BF16 = False


def _r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32) if EMULATE_BF16 else t


def _lin(x, w, b):
    return _r(F.linear(x, _r(w), b))


# --------------------------------------------------------------------------------------
# masks / positional table
# --------------------------------------------------------------------------------------
def look_ahead_mask(x: torch.Tensor, pad: int) -> torch.Tensor:
    """mask[b,0,i,j] = (x[b,j] == pad) or (j > i)          utils.py:58-83, 183-188"""
    B, L = x.shape
    j = torch.arange(L)
    future = j[None, :] > j[:, None]                       # [L,L]
    return (x == pad)[:, None, None, :] | future[None, None]


def sinusoid_table(max_seq: int, d: int) -> torch.Tensor:
    """PE[pos,i] = sin(pos / 10000^((i - i%2)/d) + (pi/2)(i%2)), built in fp64.   layers.py:9-19,22-39"""
    pos = torch.arange(max_seq, dtype=torch.float64)[:, None]
    i = torch.arange(d, dtype=torch.float64)[None, :]
    par = torch.remainder(i, 2)
    ang = pos * torch.exp(-math.log(10000.0) * i / d) * torch.exp(math.log(10000.0) / d * par) + 0.5 * math.pi * par
    return torch.sin(ang)                                   # float64 [max_seq, d]


# --------------------------------------------------------------------------------------
# relative global attention
# --------------------------------------------------------------------------------------
def srel_from_qe(q: torch.Tensor, E: torch.Tensor, len_k: int) -> torch.Tensor:
    """Srel[b,h,i,j] = q_i . E[M-1-(i-j)] for j<=i, 0 for j>i.
    (einsum + _qe_masking + _skewing, layers.py:89-92,111-133; E is sliced E[M-L:] so the
    index into the *slice* is Lq-1-(i-j).)"""
    B, H, Lq, dh = q.shape
    M = E.shape[0]
    Es = E[max(0, M - Lq):]                                 # [Lq', dh]   layers.py:111-114
    i = torch.arange(Lq)[:, None]
    j = torch.arange(len_k)[None, :]
    rel = Es.shape[0] - 1 - (i - j)                         # [Lq, Lk]
    valid = (j <= i) & (rel >= 0) & (rel < Es.shape[0])
    relc = rel.clamp(0, Es.shape[0] - 1)
    Eg = Es[relc]                                           # [Lq, Lk, dh]
    srel = torch.einsum("bhid,ijd->bhij", q, Eg)
    return srel * valid.to(q.dtype)


def attn_core(qkv: torch.Tensor, E: torch.Tensor, mask: Optional[torch.Tensor], h: int):
    """The part of RGA.forward between the q/k/v projections and `fc` (layers.py:86-106), on a fused
    qkv tensor [B,L,3d] (columns q|k|v, head hd at hd*dh..).  Returns (ctx [B,L,d], weights, logits)."""
    B, L, d3 = qkv.shape
    d = d3 // 3
    dh = d // h

    def heads(t):
        return t.reshape(B, L, h, dh).permute(0, 2, 1, 3)

    q, k, v = heads(qkv[..., :d]), heads(qkv[..., d:2 * d]), heads(qkv[..., 2 * d:])
    logits = (q @ k.transpose(-1, -2) + srel_from_qe(q, _r(E), L)) / math.sqrt(dh)
    if mask is not None:
        logits = logits + (mask.to(torch.int64) * -1e9).to(logits.dtype)
    w = torch.softmax(logits, -1)
    return _r((w @ v).permute(0, 2, 1, 3).reshape(B, L, d)), w, logits


def rga_forward(p: Params, prefix: str, x: torch.Tensor, mask: Optional[torch.Tensor], h: int
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """RelativeGlobalAttention.forward with q=k=v=x.           layers.py:64-109
    returns (out [B,L,d], attention_weights [B,h,L,L])"""
    B, L, d = x.shape
    dh = d // h

    def heads(t):
        return t.reshape(B, L, h, dh).permute(0, 2, 1, 3)

    q = heads(_lin(x, p[prefix + "Wq.weight"], p[prefix + "Wq.bias"]))
    k = heads(_lin(x, p[prefix + "Wk.weight"], p[prefix + "Wk.bias"]))
    v = heads(_lin(x, p[prefix + "Wv.weight"], p[prefix + "Wv.bias"]))
    srel = srel_from_qe(q, _r(p[prefix + "E"]), L)
    logits = (q @ k.transpose(-1, -2) + srel) / math.sqrt(dh)
    if mask is not None:
        logits = logits + (mask.to(torch.int64) * -1e9).to(logits.dtype)   # layers.py:99-100
    w = torch.softmax(logits, -1)
    ctx = _r((w @ v).permute(0, 2, 1, 3).reshape(B, L, d))
    return _lin(ctx, p[prefix + "fc.weight"], p[prefix + "fc.bias"]), w


# --------------------------------------------------------------------------------------
# encoder layer / model
# --------------------------------------------------------------------------------------
def encoder_layer(p: Params, pre: str, x: torch.Tensor, mask, h: int, rate: float, training: bool):
    """EncoderLayer.forward (post-LN, eps 1e-6, FFN d -> d/2 -> d, ReLU).   layers.py:152-161"""
    d = x.shape[-1]
    a, w = rga_forward(p, pre + "rga.", x, mask, h)
    a = F.dropout(a, rate, training)
    o1 = _r(F.layer_norm(a + x, (d,), p[pre + "layernorm1.weight"], p[pre + "layernorm1.bias"], 1e-6))
    f = F.relu(_lin(o1, p[pre + "FFN_pre.weight"], p[pre + "FFN_pre.bias"]))
    f = _lin(f, p[pre + "FFN_suf.weight"], p[pre + "FFN_suf.bias"])
    f = F.dropout(f, rate, training)
    o2 = _r(F.layer_norm(o1 + f, (d,), p[pre + "layernorm2.weight"], p[pre + "layernorm2.bias"], 1e-6))
    return o2, w


def num_layers_of(p: Params) -> int:
    n = 0
    while f"Decoder.enc_layers.{n}.rga.E" in p:
        n += 1
    return n


def decoder_stack(p: Params, x: torch.Tensor, mask, rate: float = 0.0, training: bool = False):
    """Encoder.forward: emb*sqrt(d) + PE, dropout, N layers.      layers.py:223-233"""
    emb = p["Decoder.embedding.weight"]
    d = emb.shape[1]
    h = d // 64                                             # layers.py:219
    L = x.shape[1]
    hcur = emb[x.long()] * math.sqrt(d)
    hcur = _r(hcur + sinusoid_table(L, d).to(hcur.dtype)[None])
    hcur = F.dropout(hcur, rate, training)
    ws = []
    for li in range(num_layers_of(p)):
        hcur, w = encoder_layer(p, f"Decoder.enc_layers.{li}.", hcur, mask, h, rate, training)
        ws.append(w)
    return hcur, ws


def model_forward(p: Params, x: torch.Tensor, pad: int, rate: float = 0.0, training: bool = False,
                  causal: bool = True):
    """MusicTransformer.forward (train/eval branch).          network.py:35-40
    causal=False restates generate()'s mask=None call (network.py:60-62)."""
    mask = look_ahead_mask(x, pad) if causal else None
    hcur, ws = decoder_stack(p, x, mask, rate, training)
    return _lin(hcur, p["fc.weight"], p["fc.bias"]), ws


# --------------------------------------------------------------------------------------
# loss / metrics / schedule
# --------------------------------------------------------------------------------------
def smooth_ce(logits: torch.Tensor, target: torch.Tensor, eps: float, vocab: int, pad: int) -> torch.Tensor:
    """SmoothCrossEntropyLoss (mean over non-pad targets).       criterion.py:43-67
    closed form: sum_{y!=pad} [lse - (1-eps) x_y - (eps/V) sum_v x_v] / #non-pad"""
    x = logits.reshape(-1, vocab).float()
    t = target.reshape(-1).long()
    lse = torch.logsumexp(x, -1)
    xt = x.gather(1, t.clamp(0, vocab - 1)[:, None])[:, 0]
    per = lse - (1.0 - eps) * xt - (eps / vocab) * x.sum(-1)
    keep = t != pad
    return (per * keep).sum() / keep.sum()


def accuracy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """CategoricalAccuracy: mean over ALL positions (pad included).   metrics.py:40-52,22-29"""
    return (logits.argmax(-1).long() == target.long()).float().mean()


def bucket(logits: torch.Tensor) -> torch.Tensor:
    """LogitsBucketting.                                          metrics.py:55-60"""
    return logits.argmax(-1).flatten().to(torch.int32)


def schedule_rate(step: int, d_model: int, warmup: int = 4000) -> float:
    """CustomSchedule.rate.                                       criterion.py:89-96"""
    return d_model ** -0.5 * min(step ** -0.5, step * warmup ** -1.5)


# --------------------------------------------------------------------------------------
# reference-style init + whole training step (used as the CPU baseline in bench.py)
# --------------------------------------------------------------------------------------
def init_params(vocab: int, d: int, nl: int, max_seq: int, seed: int = 0) -> Params:
    """Reference-style initialisation (E, embedding ~ N(0,1); Linear kaiming-uniform(a=sqrt5);
    LN 1/0), keyed by the reference's state_dict names."""
    g = torch.Generator().manual_seed(seed)

    def lin(o, i):
        bound = 1.0 / math.sqrt(i)
        w = (torch.rand(o, i, generator=g) * 2 - 1) * bound
        b = (torch.rand(o, generator=g) * 2 - 1) * bound
        return w, b

    p: Params = {"Decoder.embedding.weight": torch.randn(vocab, d, generator=g)}
    for li in range(nl):
        pre = f"Decoder.enc_layers.{li}."
        for nm in ("Wq", "Wk", "Wv", "fc"):
            p[pre + f"rga.{nm}.weight"], p[pre + f"rga.{nm}.bias"] = lin(d, d)
        p[pre + "rga.E"] = torch.randn(max_seq, 64, generator=g)
        p[pre + "FFN_pre.weight"], p[pre + "FFN_pre.bias"] = lin(d // 2, d)
        p[pre + "FFN_suf.weight"], p[pre + "FFN_suf.bias"] = lin(d, d // 2)
        for k in ("layernorm1", "layernorm2"):
            p[pre + k + ".weight"] = torch.ones(d)
            p[pre + k + ".bias"] = torch.zeros(d)
    p["fc.weight"], p["fc.bias"] = lin(vocab, d)
    return p


class CpuTrainer:
    """fwd + smoothed CE + accuracy + bwd + Adam(0.9,0.98,1e-9) + Noam schedule, eager CPU fp32.
    Restates the body of the loop at train.py:255-277 (the reference's hot loop)."""

    def __init__(self, p: Params, pad: int, d_cfg: int, dropout: float = 0.2, eps_ls: float = 0.1,
                 accum: int = 1):
        self.p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        self.pad, self.dropout, self.eps_ls, self.accum = pad, dropout, eps_ls, accum
        self.vocab = p["fc.weight"].shape[0]
        self.opt = torch.optim.Adam(list(self.p.values()), lr=0.0, betas=(0.9, 0.98), eps=1e-9)
        self.d_cfg, self.sched_step, self.micro = d_cfg, 0, 0

    def step(self, x: torch.Tensor, y: torch.Tensor):
        logits, _ = model_forward(self.p, x, self.pad, self.dropout, True)
        loss = smooth_ce(logits, y, self.eps_ls, self.vocab, self.pad)
        acc = accuracy(logits, y)
        (loss / self.accum).backward()
        self.micro += 1
        if self.micro % self.accum == 0:
            self.sched_step += 1
            lr = schedule_rate(self.sched_step, self.d_cfg)
            for g in self.opt.param_groups:
                g["lr"] = lr
            self.opt.step()
            self.opt.zero_grad()
        return float(loss.detach()), float(acc)


# --------------------------------------------------------------------------------------
# Event_Melody_RNN (GRU LM) single step      Event_MelodyRNN/network.py:51-61,98-104
# --------------------------------------------------------------------------------------
def gru_init_hidden(p: Params, init: torch.Tensor, layers: int, hidden: int) -> torch.Tensor:
    out = torch.tanh(F.linear(init, p["inithid_fc.weight"], p["inithid_fc.bias"]))
    return out.view(layers, init.shape[0], hidden)


def gru_step(p: Params, event: torch.Tensor, hid: torch.Tensor):
    """event int64 [1,B], hid [layers,B,H] -> (logits [1,B,V], hid').  Gate order r,z,n (torch GRU)."""
    x = p["event_embedding.weight"][event[0]]
    new = []
    for l in range(hid.shape[0]):
        gi = F.linear(x, p[f"rnn.weight_ih_l{l}"], p[f"rnn.bias_ih_l{l}"])
        gh = F.linear(hid[l], p[f"rnn.weight_hh_l{l}"], p[f"rnn.bias_hh_l{l}"])
        ir, iz, inn = gi.chunk(3, -1)
        hr, hz, hn = gh.chunk(3, -1)
        r = torch.sigmoid(ir + hr)
        z = torch.sigmoid(iz + hz)
        n = torch.tanh(inn + r * hn)
        x = (1 - z) * n + z * hid[l]
        new.append(x)
    return F.linear(x, p["output_fc.weight"], p["output_fc.bias"])[None], torch.stack(new)


def gru_train_logits(p: Params, init: torch.Tensor, events: torch.Tensor, layers: int, hidden: int,
                     primary_event: int) -> torch.Tensor:
    """Event_Melody_RNN.Train (network.py:109-116 -> SeqForward :63-84): the primary event as step 0, then the
    teacher-forced ``events`` int64 [T,B], through the GRU from ``init_to_hidden(init)``; logits [T+1,B,V].
    Dropout between layers is off (eval / dropout=0), as in the golden fixture G8."""
    hid = gru_init_hidden(p, init, layers, hidden)
    B = init.shape[0]
    ev = torch.full((1, B), primary_event, dtype=torch.long)
    outs = []
    o, hid = gru_step(p, ev, hid)
    outs.append(o)
    for t in range(events.shape[0]):
        o, hid = gru_step(p, events[t:t + 1].long(), hid)
        outs.append(o)
    return torch.cat(outs, 0)

