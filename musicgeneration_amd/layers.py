"""Host-side mirror of the reference's ``layers.py`` (mg/model/MusicTransformer/layers.py).

Same class names, constructor arguments and ``state_dict`` keys (``Decoder.enc_layers.{i}.rga.Wq.weight``
...), so reference checkpoints load unchanged -- but the arithmetic runs in libmgx.so:

* parameters live in ONE flat fp32 buffer (``FlatStore``), gradients in one flat fp32 buffer the
  backward kernels accumulate into directly (no per-parameter autograd accumulation), and a flat bf16
  shadow is what the GEMM / attention kernels read.  Wq|Wk|Wv are adjacent, so the fused
  ``[3d, d]`` QKV weight is a zero-copy view (one GEMM instead of the reference's three,
  layers.py:71-84).
* a layer's gradient segment is contiguous => it is the data-parallel all-reduce bucket
  (``musicgeneration_amd.dp``), launched as soon as the layer's backward has finished.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops

ALIGN = 64  # elements; keeps every fp32/bf16 view 128-byte aligned


def sinusoid(max_seq: int, embedding_dim: int) -> np.ndarray:
    """float64 [1, max_seq, d] table; same values as the reference's nested loops (layers.py:9-19)."""
    pos = np.arange(max_seq, dtype=np.float64)[:, None]
    i = np.arange(embedding_dim, dtype=np.float64)[None, :]
    par = np.mod(i, 2)
    ang = pos * np.exp(-math.log(10000) * i / embedding_dim) * np.exp(math.log(10000) / embedding_dim * par) \
        + 0.5 * math.pi * par
    return np.sin(ang)[None]


class DynamicPositionEmbedding(torch.nn.Module):
    """layers.py:22-39.  The table is built once in float64 and kept device-resident in fp32
    (the reference re-uploads the fp64 table on every forward)."""

    def __init__(self, embedding_dim, max_seq=2048):
        super().__init__()
        self.positional_embedding = sinusoid(max_seq, embedding_dim)
        self.register_buffer("_pe", torch.from_numpy(self.positional_embedding[0]).to(torch.float32),
                             persistent=False)

    def table(self) -> torch.Tensor:
        return self._pe

    def forward(self, x):   # API parity; the fused path uses table() inside embed_pe
        return x + self._pe[: x.size(1)].to(x.dtype)[None]


def key_padding_from_mask(mask: torch.Tensor, L: int):
    """The kernels build the reference's look-ahead mask themselves (future keys, structurally; padded keys, from a
    bitmap).  A mask tensor handed to a layer (utils.get_masked_with_pad_tensor's ``[B,1,L,L]``, nonzero = masked) must
    therefore BE such a mask: returns the padded-key flags ``[B,L]`` (read off the last query row, which sees every key)
    after checking that ``mask == padded[key] | (key > query)``.  Anything else raises."""
    m = mask
    if m.dim() == 4:
        if m.shape[1] != 1:
            raise ValueError("per-head masks are not supported (expected [B,1,L,L])")
        m = m[:, 0]
    if m.dim() != 3 or m.shape[-1] != L or m.shape[-2] != L:
        raise ValueError(f"mask must be [B,1,{L},{L}] (got {tuple(mask.shape)})")
    m = m != 0
    padded = m[:, L - 1, :]
    future = torch.ones(L, L, dtype=torch.bool, device=m.device).triu(1)
    if not torch.equal(m, padded[:, None, :] | future[None]):
        raise ValueError("the MI355X attention kernels implement causal attention with key padding only: the mask must be "
                         "the look-ahead mask of utils.get_masked_with_pad_tensor (padded keys | future keys)")
    return padded


def _pad_bitmap_from_flags(padded: torch.Tensor) -> torch.Tensor:
    """bool [B,L] -> int32 [B,L/32] in mgx_pad_bitmap's layout (bit j&31 of word j>>5)"""
    B, L = padded.shape
    w = (padded.view(B, L // 32, 32).to(torch.int64) << torch.arange(32, device=padded.device, dtype=torch.int64)).sum(-1)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w)
    return w.to(torch.int32).contiguous()


def _need_mask(mask):
    """mask=None is the reference's sampling call (network.py:60-62): bidirectional attention, relative term for j <= i only.
    The kernels run it (mgx_rel_attn_fwd_nomask) for inference; there is no backward for it."""
    if mask is None and torch.is_grad_enabled():
        raise NotImplementedError(
            "mask=None (the bidirectional attention of the reference's generate(), network.py:60) is an inference path on the "
            "MI355X kernels: call it under torch.no_grad(), or pass the look-ahead mask for training")


class RelativeGlobalAttention(torch.nn.Module):
    """layers.py:47-133.  Holds the reference's parameters under the reference's names.  ``forward([q,k,v], mask)`` runs the
    fused relative attention kernels on its own (projections -> mgx_rel_attn_fwd -> fc), with ordinary autograd gradients;
    inside MusicTransformer the whole encoder block runs as one node instead (ops._EncoderLayer)."""

    def __init__(self, h=4, d=256, add_emb=False, max_seq=2048, **kwargs):
        super().__init__()
        if d // h != 64:
            raise ValueError("the MI355X kernels fix dh = d/h = 64 (the reference always uses h = d//64, layers.py:219)")
        self.len_k = None
        self.max_seq = max_seq
        self.h = h
        self.d = d
        self.dh = d // h
        self.Wq = torch.nn.Linear(self.d, self.d)
        self.Wk = torch.nn.Linear(self.d, self.d)
        self.Wv = torch.nn.Linear(self.d, self.d)
        self.fc = torch.nn.Linear(d, d)
        self.additional = add_emb
        self.E = torch.nn.Parameter(torch.randn([self.max_seq, int(self.dh)]))
        self.need_weights = True        # the reference always returns the [B,h,L,L] weights; set False to skip that pass

    def forward(self, inputs, mask=None, **kwargs):
        """inputs = [q, k, v], each [B,L,d] (layers.py:64-109) -> (out [B,L,d], attention_weights [B,h,L,L] or None).
        Self-attention shapes only (len_q == len_k <= max_seq; any such L since round 6: other lengths than multiples of the
        kernels' 32-key tile are zero-padded at the end, the padded keys masked, the padded rows dropped).  Computes in bf16;
        returns q's dtype."""
        q, k, v = inputs
        if not (q.shape == k.shape == v.shape) or q.dim() != 3 or q.shape[-1] != self.d:
            raise ValueError("expected three [B,L,d] tensors of one shape")
        _need_mask(mask)
        B, L, _ = q.shape
        if L < 1 or L > self.max_seq:
            raise ValueError(f"sequence length {L} must be in 1 .. max_seq={self.max_seq}")
        self.len_q = self.len_k = L
        Lp = (L + 31) // 32 * 32
        padded = None if mask is None else key_padding_from_mask(mask, L)
        same = q is k and k is v
        E = self.E
        if Lp != L:
            tail = (0, 0, 0, Lp - L)
            q = torch.nn.functional.pad(q, tail)
            k, v = (q, q) if same else (torch.nn.functional.pad(k, tail), torch.nn.functional.pad(v, tail))
            if padded is not None:
                padded = torch.nn.functional.pad(padded, (0, Lp - L), value=True)
            if Lp > self.max_seq:                                # distances >= max_seq only occur for padded queries: zero rows in FRONT
                E = torch.cat([torch.zeros(Lp - self.max_seq, E.shape[1], dtype=E.dtype, device=E.device), E], 0)
        padbits = None if padded is None else _pad_bitmap_from_flags(padded)
        bf = torch.bfloat16
        if same:
            x16 = q.to(bf).contiguous()
            qkv = ops.linear_std(x16, torch.cat([self.Wq.weight, self.Wk.weight, self.Wv.weight], 0),
                                 torch.cat([self.Wq.bias, self.Wk.bias, self.Wv.bias], 0))
        else:
            qkv = torch.cat([ops.linear_std(t.to(bf).contiguous(), lin.weight, lin.bias)
                             for t, lin in ((q, self.Wq), (k, self.Wk), (v, self.Wv))], -1)
        if mask is None:                                          # the reference's sampling call: no look-ahead, no padding mask
            att = ops.rel_attn_fwd_nomask(qkv, E.detach().to(bf).contiguous(), L)      # keys >= L do not exist
            return ops.linear_std(att[:, :L].contiguous(), self.fc.weight, self.fc.bias).to(inputs[0].dtype), None
        att, lse = ops.rel_attn_std(qkv, E, padbits)               # E is cut to its last Lp rows inside (M >= Lp)
        out = ops.linear_std(att[:, :L].contiguous() if Lp != L else att, self.fc.weight, self.fc.bias)
        weights = None
        if self.need_weights:
            with torch.no_grad():
                weights = ops.rel_attn_weights(qkv.detach(), E.detach().to(bf).contiguous(), padbits, lse)
                if Lp != L:
                    weights = weights[:, :, :L, :L]
        return out.to(inputs[0].dtype), weights


class EncoderLayer(torch.nn.Module):
    """layers.py:137-161: post-LN block, FFN d -> d/2 -> d with ReLU, LayerNorm eps 1e-6."""

    def __init__(self, d_model, rate=0.1, h=16, additional=False, max_seq=2048):
        super().__init__()
        self.d_model = d_model
        self.rate = rate
        self.rga = RelativeGlobalAttention(h=h, d=d_model, max_seq=max_seq, add_emb=additional)
        self.FFN_pre = torch.nn.Linear(self.d_model, self.d_model // 2)
        self.FFN_suf = torch.nn.Linear(self.d_model // 2, self.d_model)
        self.layernorm1 = torch.nn.LayerNorm(self.d_model, eps=1e-6)
        self.layernorm2 = torch.nn.LayerNorm(self.d_model, eps=1e-6)
        self.dropout1 = torch.nn.Dropout(rate)
        self.dropout2 = torch.nn.Dropout(rate)
        self._seed_ctr = 0

    def forward(self, x, mask=None, **kwargs):
        """layers.py:152-161 on its own (see RelativeGlobalAttention.forward): -> (out2, attention weights)"""
        attn_out, w = self.rga([x, x, x], mask)
        bf = torch.bfloat16
        p = self.rate if self.training else 0.0
        self._seed_ctr += 1
        seed = (torch.initial_seed() * 1000003 + self._seed_ctr * 64) & 0x7FFFFFFFFFFFFFFF
        # dropout is fused into the residual + LayerNorm kernel (mask = pure function of (seed, element index))
        out1 = ops.add_ln_std(attn_out.to(bf).contiguous(), x.to(bf).contiguous(), self.layernorm1.weight, self.layernorm1.bias,
                              self.layernorm1.eps, p, seed + 1)
        ffn = ops.linear_std(out1, self.FFN_pre.weight, self.FFN_pre.bias, act=1)
        ffn = ops.linear_std(ffn, self.FFN_suf.weight, self.FFN_suf.bias)
        out2 = ops.add_ln_std(ffn, out1, self.layernorm2.weight, self.layernorm2.bias, self.layernorm2.eps, p, seed + 2)
        return out2.to(x.dtype), w

    # order of this layer's parameters inside the flat buffers (Wq|Wk|Wv adjacent!)
    FLAT_ORDER = ["rga.Wq.weight", "rga.Wk.weight", "rga.Wv.weight", "rga.Wq.bias", "rga.Wk.bias", "rga.Wv.bias",
                  "rga.fc.weight", "rga.fc.bias", "rga.E", "layernorm1.weight", "layernorm1.bias",
                  "FFN_pre.weight", "FFN_pre.bias", "FFN_suf.weight", "FFN_suf.bias",
                  "layernorm2.weight", "layernorm2.bias"]


class FlatStore:
    """Flat fp32 parameter / gradient buffers + bf16 shadow for an ordered list of parameters."""

    def __init__(self, named: List[Tuple[str, torch.nn.Parameter]], device, buckets: List[Tuple[str, List[str]]],
                 padded: Optional[Dict[str, int]] = None, colpad: Optional[Dict[str, int]] = None):
        """``padded[name]`` = number of elements to reserve for ``name`` (>= numel): the tail stays zero in
        every buffer (parameters, gradients, moments, shadow) -- used to pad the vocabulary projection to a
        multiple of 64 rows so all GEMM dimensions are tile-aligned.
        ``colpad[name]`` = padded COLUMN count of a 2-D parameter: it is stored as [rows, colpad] with zero columns
        beyond its own, and the Parameter becomes the strided view ``[:, :cols]`` of that -- the FFN's second
        weight when d_model / 2 is not a multiple of 64 (the GEMMs' reduction length; round 6)."""
        self.names = [n for n, _ in named]
        self.offsets: Dict[str, Tuple[int, int]] = {}
        self.reserved: Dict[str, int] = {}
        self.colpad: Dict[str, int] = dict(colpad or {})
        padded = padded or {}
        off = 0
        for n, p in named:
            self.offsets[n] = (off, p.numel())
            res = max(p.numel(), padded.get(n, 0))
            if n in self.colpad:
                assert p.dim() == 2 and self.colpad[n] >= p.shape[1]
                res = p.shape[0] * self.colpad[n]
            self.reserved[n] = res
            off += (res + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.param = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=device)
        self.params = {}
        self.shapes = {n: tuple(p.shape) for n, p in named}
        for n, p in named:
            view = self.view(self.param, n)
            view.copy_(p.data)
            p.data = view
            p.grad = self.view(self.grad, n)
            self.params[n] = p
        # bucket = (name, start, end) contiguous range of the flat buffers
        self.buckets: List[Tuple[str, int, int]] = []
        for bname, members in buckets:
            lo = min(self.offsets[m][0] for m in members)
            hi = max(self.offsets[m][0] + (self.reserved[m] + ALIGN - 1) // ALIGN * ALIGN for m in members)
            self.buckets.append((bname, lo, hi))
        self._synced_version = -1
        self.sync_shadow(force=True)

    # ---- views ------------------------------------------------------------------------------------
    def view(self, buf, name):
        """the region of parameter ``name`` in flat buffer ``buf`` (parameters, gradients, shadow, or an optimiser's moments laid
        out alike) with the parameter's own shape: contiguous, or -- column-padded parameters -- the [:, :cols] view of its
        [rows, colpad] storage"""
        o, k = self.offsets[name]
        shp = self.shapes[name]
        if name in self.colpad:
            cp = self.colpad[name]
            return buf[o:o + shp[0] * cp].view(shp[0], cp)[:, :shp[1]]
        return buf[o:o + k].view(shp)

    def w(self, name):          # bf16 shadow view with the parameter's shape
        return self.view(self.shadow, name)

    def g(self, name):          # fp32 gradient view
        return self.view(self.grad, name)

    def padded_view(self, name, rows, cols=None, what="shadow"):
        """view over the reserved (zero-padded) region of ``name`` as [rows, cols] (or [rows])"""
        o = self.offsets[name][0]
        buf = {"shadow": self.shadow, "grad": self.grad, "param": self.param}[what]
        n = rows * (cols or 1)
        assert n <= self.reserved[name]
        v = buf[o:o + n]
        return v.view(rows, cols) if cols else v

    def fused(self, first, last, rows, cols, what="shadow"):
        """one view over adjacent parameters first..last (e.g. Wq|Wk|Wv -> [3d, d])"""
        o = self.offsets[first][0]
        buf = {"shadow": self.shadow, "grad": self.grad, "param": self.param}[what]
        end = self.offsets[last][0] + self.offsets[last][1]
        assert end - o == rows * cols, "fused view needs adjacent, unpadded parameters"
        return buf[o:end].view(rows, cols)

    # ---- shadow / grad maintenance ------------------------------------------------------------------
    def sync_shadow(self, force=False):
        """Re-round the bf16 shadow if the fp32 parameters were modified by anything but our Adam kernel
        (torch in-place ops bump the version counter of the parameter they touch)."""
        v = self._versions()
        if force or v != self._synced_version:
            ops.cast_bf16(self.param, self.shadow)
            self._synced_version = self._versions()

    def _versions(self):
        # ``p.data = view`` keeps the Parameter's OWN version counter (it is not shared with the flat base),
        # so in-place updates by torch optimizers / load_state_dict show up per parameter.
        return self.param._version + sum(p._version for p in self.params.values())

    def attach_grads(self):
        """``optimizer.zero_grad(set_to_none=True)`` drops our views: re-attach (and zero) them."""
        lost = False
        for n, p in self.params.items():
            o, k = self.offsets[n]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                lost = True
                break
        if lost:
            self.grad.zero_()
            for n, p in self.params.items():
                p.grad = self.view(self.grad, n)


class Encoder(torch.nn.Module):
    """layers.py:208-233.  ``forward`` runs the whole stack through the HIP kernels."""

    def __init__(self, num_layers, d_model, input_vocab_size, rate=0.1, max_len=None):
        super().__init__()
        self.d_model = d_model
        self.num_layers = num_layers
        self.rate = rate
        self.max_len = max_len
        self.embedding = torch.nn.Embedding(num_embeddings=input_vocab_size, embedding_dim=d_model)
        self.pos_encoding = DynamicPositionEmbedding(self.d_model, max_seq=max_len)
        self.enc_layers = torch.nn.ModuleList(
            [EncoderLayer(d_model, rate, h=self.d_model // 64, additional=False, max_seq=max_len)
             for _ in range(num_layers)])
        self.dropout = torch.nn.Dropout(rate)
        self._seed_ctr = 0

    def forward(self, x, mask=None):
        """layers.py:223-233 on its own: tokens [B,L] -> (hidden [B,L,d] fp32, [attention weights per layer]).
        MusicTransformer.forward does NOT go through here (it runs the fused flat-buffer path, network._logits)."""
        _need_mask(mask)
        weights = []
        tok = x.to(torch.int32).contiguous()
        p = self.rate if self.training else 0.0
        self._seed_ctr += 1
        seed = (torch.initial_seed() * 1000003 + self._seed_ctr * 64) & 0x7FFFFFFFFFFFFFFF
        h = _EmbedStd.apply(tok, self.embedding.weight, self.pos_encoding.table()[: tok.shape[1]].contiguous(), p, seed)
        for layer in self.enc_layers:
            h, w = layer(h, mask)
            weights.append(w)
        return h.float(), weights


class _EmbedStd(torch.autograd.Function):
    """embedding * sqrt(d) + PE (+dropout) with an ordinary returned gradient (stand-alone Encoder.forward)"""

    @staticmethod
    def forward(ctx, tok, table, pe, p_drop, seed):
        ctx.save_for_backward(tok)
        ctx.meta = (table.shape, table.dtype, p_drop, seed)
        return ops.embed_pe_fwd(tok, table.detach().float().contiguous(), pe, p_drop, seed)

    @staticmethod
    def backward(ctx, dout):
        (tok,) = ctx.saved_tensors
        shape, dt, p_drop, seed = ctx.meta
        g = torch.zeros(shape, dtype=torch.float32, device=dout.device)
        ops.embed_bwd(tok, dout.contiguous(), g, p_drop, seed)
        return None, g.to(dt), None, None, None
