"""Host-side mirror of the reference's ``network.py`` (mg/model/MusicTransformer/network.py:14-84).

``MusicTransformer`` keeps the reference's constructor, attributes, ``state_dict`` keys and the
three-way ``forward`` convention (train -> logits; eval -> (logits, weights); after ``test()`` ->
``generate(...).tolist()``), and runs on the HIP kernels of libmgx.so.  There is no eager/CPU
fallback: ``forward`` on a CPU tensor raises.
"""
from __future__ import annotations

import os

from typing import Optional

import numpy as np
import torch

from . import config, ops
from .layers import Encoder, EncoderLayer, FlatStore


class MusicTransformer(torch.nn.Module):
    def __init__(self, embedding_dim=256, vocab_size=388 + 2, num_layer=6,
                 max_seq=2048, dropout=0.2, debug=False, loader_path=None, dist=False, writer=None):
        super().__init__()
        self.infer = False
        if loader_path is not None:
            raise NotImplementedError("loader_path: the reference calls an undefined load_config_file "
                                      "(network.py:19-20); pass the hyper-parameters explicitly")
        self._debug = debug
        self.max_seq = max_seq
        self.num_layer = num_layer
        self.embedding_dim = embedding_dim
        self.vocab_size = vocab_size
        self.dist = dist
        self.writer = writer
        self.dropout_rate = dropout
        # the reference reads the pad id from the global config module (network.py:37); the default
        # is the same rule (pad = last vocabulary id), overridable per model
        self.pad_token = vocab_size - 1
        self.vocab_padded = (vocab_size + 63) // 64 * 64     # rows of the vocabulary GEMM (zero rows beyond V)
        if embedding_dim % 64 != 0:
            raise ValueError("the MI355X kernels fix the head width at 64 (the reference's h = d // 64, dh = d // h, layers.py:219): "
                             f"embedding_dim must be a multiple of 64, got {embedding_dim}")
        self.ffn_padded = (embedding_dim // 2 + 63) // 64 * 64      # FFN width in the flat buffers (zero rows / columns beyond d/2)
        self.Decoder = Encoder(num_layers=self.num_layer, d_model=self.embedding_dim,
                               input_vocab_size=self.vocab_size, rate=dropout, max_len=max_seq)
        self.fc = torch.nn.Linear(self.embedding_dim, self.vocab_size)
        self.return_attention_weights = False   # eval-mode [B,h,L,L] weights are a debug output
        self._store: Optional[FlatStore] = None
        self._seed_ctr = 0
        self._dp = None                          # set by dp.DataParallel
        self._pad_flag: Optional[torch.Tensor] = None   # device int32[1], sticky: a row started with padding and held real tokens

    # ------------------------------------------------------------------------------------------
    # flat storage
    # ------------------------------------------------------------------------------------------
    def _flat_order(self):
        named = dict(self.named_parameters())
        order = ["Decoder.embedding.weight"]
        buckets = [("embedding", ["Decoder.embedding.weight"])]
        for i in range(self.num_layer):
            names = [f"Decoder.enc_layers.{i}.{n}" for n in EncoderLayer.FLAT_ORDER]
            order += names
            buckets.append((f"layer{i}", names))
        order += ["fc.weight", "fc.bias"]
        buckets.append(("fc", ["fc.weight", "fc.bias"]))
        assert set(order) == set(named), "flat order must cover every parameter exactly once"
        padded = {"fc.weight": self.vocab_padded * self.embedding_dim, "fc.bias": self.vocab_padded}
        # FFN width d/2 (layers.py:143-144) padded to the GEMMs' reduction granule of 64 when d = 64 * odd (d = 192, 320, ...): zero
        # rows / bias entries of FFN_pre, zero columns of FFN_suf -- ReLU(0) = 0 feeds zero columns, every gradient there is exactly 0,
        # Adam leaves zeros zero.  The Parameters keep the reference's shapes (state_dict, checkpoints).
        colpad = {}
        H, Hp, d = self.embedding_dim // 2, self.ffn_padded, self.embedding_dim
        if Hp != H:
            for i in range(self.num_layer):
                pre = f"Decoder.enc_layers.{i}."
                padded[pre + "FFN_pre.weight"] = Hp * d
                padded[pre + "FFN_pre.bias"] = Hp
                colpad[pre + "FFN_suf.weight"] = Hp
        return [(n, named[n]) for n in order], buckets, padded, colpad

    def store(self) -> FlatStore:
        dev = self.fc.weight.device
        if self._store is None or self._store.param.device != dev:
            if dev.type != "cuda":
                raise ops._lib.MgxError("MusicTransformer runs on the MI355X kernels only: move it to a HIP "
                                        "device first (model.to('cuda')); there is no CPU fallback")
            named, buckets, padded, colpad = self._flat_order()
            self._store = FlatStore(named, dev, buckets, padded, colpad)
            if os.environ.get("MGX_DETERMINISTIC", "0") == "1" and not ops.deterministic():
                ops.set_deterministic(True, dev)
        return self._store

    def _apply(self, fn, *a, **k):     # .to()/.cuda() re-materialise parameters: rebuild lazily
        self._store = None
        return super()._apply(fn, *a, **k)

    def _next_seed(self) -> int:
        # under data parallelism every rank usually calls torch.manual_seed with the same value: the rank enters the seed, or
        # all ranks would draw the same dropout masks for their (different) rows
        self._seed_ctr += 1
        rank = self._dp.rank if self._dp is not None else 0
        return ((torch.initial_seed() + 0x9E3779B9 * rank) * 1000003 + self._seed_ctr * 64) & 0x7FFFFFFFFFFFFFFF

    # ------------------------------------------------------------------------------------------
    # the hot path: tokens -> logits                 network.py:37-39 + layers.py:223-233,152-161
    # ------------------------------------------------------------------------------------------
    def _logits(self, x: torch.Tensor, wsink: Optional[list] = None) -> torch.Tensor:
        st = self.store()
        st.sync_shadow()
        training = self.training and torch.is_grad_enabled()
        if training:
            st.attach_grads()
        B, L = x.shape
        if L < 1 or L > self.max_seq:
            raise ValueError(f"sequence length {L} must be in 1 .. max_seq={self.max_seq}")
        tok = x.to(torch.int32).contiguous()
        # The kernels sweep 32-key tiles.  Any other length (the reference takes every L <= max_seq, layers.py:64-109) is
        # right-padded with pad tokens up to the next multiple of 32: trailing pads are masked keys and lie in the causal future
        # of every real position, so the real rows' logits are the reference's; the padded rows are sliced off below and carry
        # no gradient.
        Lp = (L + 31) // 32 * 32
        if Lp != L:
            tok = torch.cat([tok, torch.full((B, Lp - L), self.pad_token, dtype=torch.int32, device=tok.device)], 1)
        d = self.embedding_dim
        p = self.dropout_rate if self.training else 0.0
        seed = self._next_seed()
        dp = self._dp
        done = ((lambda name: (lambda: dp.bucket_ready(name))) if (dp is not None and (dp.world > 1 or dp.force) and training)
                else (lambda name: None))

        if self._pad_flag is None or self._pad_flag.device != tok.device:
            self._pad_flag = torch.zeros(1, dtype=torch.int32, device=tok.device)
        padbits = ops.pad_bitmap(tok, self.pad_token, self._pad_flag)
        pe = self.Decoder.pos_encoding.table()
        layer_params = self._layer_params()
        if Lp > self.max_seq:
            # max_seq itself is not a multiple of 32 and L reaches into its last partial tile: the padded rows would index the
            # positional table and the relative embedding beyond max_seq.  Both get zero rows there (E at the FRONT: distance
            # delta reads E[M - 1 - delta], and only padded queries have delta >= max_seq); the padded rows' dE is exactly zero
            # (their dO is), the real rows' is folded back into the layer's gradient slot when the block's backward has run.
            extra = Lp - self.max_seq
            pe = torch.cat([pe, torch.zeros(extra, d, dtype=pe.dtype, device=pe.device)], 0)
            layer_params, done = self._params_for_padded_E(layer_params, extra, done, training)
        P = st.params
        h = ops.embed_pe(tok, P["Decoder.embedding.weight"], pe, p, seed, st.g("Decoder.embedding.weight"),
                         done("embedding"))
        for i, lp in enumerate(layer_params):
            # the layer's gradient bucket is complete when the block's backward (ending in the QKV projection) has run
            h = ops.encoder_layer(h, lp, padbits, p, seed + 4 * i, done(f"layer{i}"), wsink)
        Vp = self.vocab_padded
        logits = ops.linear(h, P["fc.weight"], st.padded_view("fc.weight", Vp, d), st.padded_view("fc.bias", Vp, None, "param"),
                            0, st.padded_view("fc.weight", Vp, d, "grad"), st.padded_view("fc.bias", Vp, None, "grad"),
                            done("fc"))
        if wsink is not None and Lp != L:
            wsink[:] = [w_[:, :, :L, :L] for w_ in wsink]
        # [B, Lp, Vp] storage, [B, L, V] view: columns >= V are exact zeros (zero weight rows, zero bias)
        return logits[:, :L, : self.vocab_size]

    def _params_for_padded_E(self, layer_params, extra, done, training):
        """per-layer operand sets whose relative embedding has `extra` zero rows in front (see _logits), and a bucket callback
        that first folds the temporary dE back into the layer's gradient slot"""
        out, folds = [], {}
        for i, lp in enumerate(layer_params):
            q = ops.LayerParams()
            for k in ops.LayerParams.__slots__:
                setattr(q, k, getattr(lp, k))
            q.E = torch.cat([torch.zeros(extra, 64, dtype=lp.E.dtype, device=lp.E.device), lp.E], 0).contiguous()
            if training:
                q.gE = torch.zeros(q.E.shape, dtype=torch.float32, device=lp.E.device)

                def fold(tmp=q.gE, slot=lp.gE):
                    tmp.record_stream(torch.cuda.current_stream())      # (the bucket callback may run on the side stream of ops.configure_streams)
                    slot.add_(tmp[extra:])
                folds[f"layer{i}"] = fold
            out.append(q)

        def done2(name):
            inner, fold = done(name), folds.get(name)
            if fold is None:
                return inner

            def both():
                fold()
                if inner is not None:
                    inner()
            return both
        return out, done2

    def _layer_params(self):
        """per-layer kernel operands as views of the flat buffers (rebuilt when the store is)"""
        st = self.store()
        if getattr(self, "_lp_store", None) is st:
            return self._lp
        d, P, out = self.embedding_dim, st.params, []
        for i in range(self.num_layer):
            pre = f"Decoder.enc_layers.{i}."
            lp = ops.LayerParams()
            lp.wqkv = st.fused(pre + "rga.Wq.weight", pre + "rga.Wv.weight", 3 * d, d)
            lp.gqkv = st.fused(pre + "rga.Wq.weight", pre + "rga.Wv.weight", 3 * d, d, "grad")
            lp.bqkv = st.fused(pre + "rga.Wq.bias", pre + "rga.Wv.bias", 1, 3 * d, "param").view(3 * d)
            lp.gbqkv = st.fused(pre + "rga.Wq.bias", pre + "rga.Wv.bias", 1, 3 * d, "grad").view(3 * d)
            lp.E, lp.gE = st.w(pre + "rga.E"), st.g(pre + "rga.E")
            # the bias gradients of `fc` and `FFN_suf` are column sums of the LayerNorm backward's dx: that
            # kernel emits them
            lp.wfc, lp.bfc = st.w(pre + "rga.fc.weight"), P[pre + "rga.fc.bias"].data
            lp.gwfc, lp.gbfc = st.g(pre + "rga.fc.weight"), st.g(pre + "rga.fc.bias")
            lp.g1, lp.b1 = P[pre + "layernorm1.weight"].data, P[pre + "layernorm1.bias"].data
            lp.gg1, lp.gb1 = st.g(pre + "layernorm1.weight"), st.g(pre + "layernorm1.bias")
            H, Hp = d // 2, self.ffn_padded
            if Hp == H:
                lp.wpre, lp.bpre = st.w(pre + "FFN_pre.weight"), P[pre + "FFN_pre.bias"].data
                lp.gwpre, lp.gbpre = st.g(pre + "FFN_pre.weight"), st.g(pre + "FFN_pre.bias")
                lp.wsuf, lp.gwsuf = st.w(pre + "FFN_suf.weight"), st.g(pre + "FFN_suf.weight")
            else:           # d = 64 * odd: the kernels see the zero-padded width (FlatStore: padded rows / colpad columns)
                lp.wpre, lp.bpre = st.padded_view(pre + "FFN_pre.weight", Hp, d), st.padded_view(pre + "FFN_pre.bias", Hp, None, "param")
                lp.gwpre, lp.gbpre = st.padded_view(pre + "FFN_pre.weight", Hp, d, "grad"), st.padded_view(pre + "FFN_pre.bias", Hp, None, "grad")
                lp.wsuf, lp.gwsuf = st.padded_view(pre + "FFN_suf.weight", d, Hp), st.padded_view(pre + "FFN_suf.weight", d, Hp, "grad")
            lp.bsuf, lp.gbsuf = P[pre + "FFN_suf.bias"].data, st.g(pre + "FFN_suf.bias")
            lp.g2, lp.b2 = P[pre + "layernorm2.weight"].data, P[pre + "layernorm2.bias"].data
            lp.gg2, lp.gb2 = st.g(pre + "layernorm2.weight"), st.g(pre + "layernorm2.bias")
            out.append(lp)
        self._lp_store, self._lp = st, out
        return out

    def check_no_leading_pads(self) -> None:
        """The library-boundary twin of ``utils.check_no_leading_pads``: every ``forward`` lets the bitmap kernel record, on the
        device, whether some row STARTED with padding and held real tokens later (rows with fully masked queries, outside the
        parity contract with the reference; trailing and interior pads are masked like the reference
        masks them and are accepted).  Reading the record is a device synchronisation, so it is done here, on request --
        train.py calls it where it prints metrics, bench.py after its timed region -- and not inside ``forward``.  Raises
        ValueError and clears the record."""
        if self._pad_flag is not None and int(self._pad_flag.item()) != 0:
            self._pad_flag.zero_()
            raise ValueError(f"a batch handed to MusicTransformer.forward had a row that starts with padding token {self.pad_token} "
                             "and holds real tokens later: leading padding is outside the parity contract with the reference "
                             "(fully masked queries); pads may trail or sit inside a sequence")

    check_pads_trail = check_no_leading_pads      # the name of rounds 4-5

    def forward(self, x, length=None, writer=None):
        if self.training or not self.infer:
            if self.training:
                return self._logits(x)
            # eval: (logits, [attention_weights per layer]) as network.py:40.  The [B,h,L,L] fp32 weights are a
            # debug output (268 MB per layer at cfg2, B=2): they are materialised only on request.
            ws = [] if self.return_attention_weights else None
            logits = self._logits(x, ws)
            return logits, (ws if ws is not None else [])
        return self.generate(x, length, None).contiguous().tolist()

    # ------------------------------------------------------------------------------------------
    # sampling                                                                   network.py:44-80
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _logits_nomask(self, window: torch.Tensor) -> torch.Tensor:
        """logits [B,W,V] of ``Decoder(window, mask=None)`` + fc -- the reference's sampling call (network.py:60-63): every
        position attends to every position of the window, the relative term only reaches back (j <= i).  The window is
        right-padded to the kernels' multiple of 32; the padding rows are excluded as keys (Lk = W) and dropped as queries."""
        st = self.store()
        st.sync_shadow()
        B, W = window.shape
        Lp = (W + 31) // 32 * 32
        if W > self.max_seq:
            raise ValueError(f"window {W} > max_seq={self.max_seq}")
        dev, d, Pm = st.param.device, self.embedding_dim, st.params
        seq = torch.zeros(B, Lp, dtype=torch.int32, device=dev)
        seq[:, :W] = window.to(torch.int32)
        pe, layer_params = self.Decoder.pos_encoding.table(), self._layer_params()
        if Lp > self.max_seq:          # max_seq is no multiple of 32: zero rows for the padded positions (see _logits)
            pe = torch.cat([pe, torch.zeros(Lp - self.max_seq, d, dtype=pe.dtype, device=pe.device)], 0)
            layer_params, _ = self._params_for_padded_E(layer_params, Lp - self.max_seq, lambda name: None, False)
        hh = ops.embed_pe_fwd(seq, Pm["Decoder.embedding.weight"].data, pe)
        for lp in layer_params:
            qkv = ops.linear_fwd(hh, lp.wqkv, lp.bqkv, 0)
            att = ops.rel_attn_fwd_nomask(qkv, lp.E, W)
            o1 = ops.add_ln_fwd(ops.linear_fwd(att, lp.wfc, lp.bfc, 0), hh, lp.g1, lp.b1, 1e-6)[0]
            f = ops.linear_fwd(ops.linear_fwd(o1, lp.wpre, lp.bpre, 1), lp.wsuf, lp.bsuf, 0)
            hh = ops.add_ln_fwd(f, o1, lp.g2, lp.b2, 1e-6)[0]
        Vp = self.vocab_padded
        logits = ops.linear_fwd(hh, st.padded_view("fc.weight", Vp, d), st.padded_view("fc.bias", Vp, None, "param"), 0)
        return logits[:, :W, : self.vocab_size]

    @torch.no_grad()
    def next_token_probs(self, window: torch.Tensor, reference_mask: bool = False) -> torch.Tensor:
        """softmax of the logits that follow the last token of ``window`` [B,W].  Default: causal semantics (the
        training-time mask, see DESIGN.md 'decode semantics'); windows that are no multiple of 32 are right-padded inside
        ``_logits`` with pad tokens, which lie in the masked future of every real position.  ``reference_mask=True``: the reference's own
        sampling call, ``Decoder(window, mask=None)`` (network.py:60-62)."""
        B, W = window.shape
        if reference_mask:
            return torch.softmax(self._logits_nomask(window)[:, W - 1].float(), -1)
        was = self.training                   # (_logits pads the window to the kernels' 32-key tile itself)
        self.eval()
        logits = self._logits(window)[:, W - 1].float()
        self.train(was)
        return torch.softmax(logits, -1)

    @torch.no_grad()
    def generate(self, prior: torch.Tensor, length=2048, tf_board_writer=None, temperature: float = 1.0,
                 top_k: int = 0, top_p: float = 1.0, reference_mask: bool = False):
        """Autoregressive sampling with the reference's sliding window (config.threshold_len) and
        full-softmax categorical sampling by default (top_k=0, top_p=1.0 == the reference's
        OneHotCategorical, network.py:73-74); top-k / top-p / temperature are opt-in extras.
        ``reference_mask=True`` reproduces the reference's step exactly -- ``Decoder(decode_array, None)``, i.e. NO look-ahead
        mask at sampling time (network.py:60) -- instead of the training-time causal semantics."""
        decode_array = prior
        result_array = prior
        for _ in range(length):
            if decode_array.size(1) >= config.threshold_len:
                decode_array = decode_array[:, 1:]
            probs = self.next_token_probs(decode_array, reference_mask)
            probs = filter_probs(probs, temperature, top_k, top_p)
            nxt = torch.multinomial(probs, 1).to(decode_array.dtype)
            decode_array = torch.cat((decode_array, nxt), dim=-1)
            result_array = torch.cat((result_array, nxt), dim=-1)
        return result_array

    # ------------------------------------------------------------------------------------------
    # KV-cache decode (cfg5): O(t) per token instead of the reference's O(W^2) recompute
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_cached(self, prior: torch.Tensor, length: int, temperature: float = 1.0, top_k: int = 0,
                        top_p: float = 1.0, seed: int = 0, use_graph: bool = True, return_probs: bool = False,
                        grammar=None, prefill: str = "auto", return_cache: bool = False, groups: Optional[int] = None,
                        masked_groups: bool = False):
        """Sample ``length`` events after ``prior`` [B,P] with per-layer K/V caches and absolute positions
        0..P+length-1 (requires P+length <= max_seq; no sliding window).  Every step runs
        embed -> N x (QKV GEMM, cached relative attention, fc, LN, FFN, LN) -> vocabulary GEMM -> fused
        sampler; the position lives on the device, so after a warm-up step the whole step is captured in
        one graph and replayed per token.  Returns int32 [B, P+length] (and, if ``return_probs``, the
        f32 [B, P+length, V] next-token distributions, position p = distribution after token p).
        ``prefill``: "batched" runs the first P-1 prior tokens through the full-sequence (training) kernels in ONE pass and
        copies every layer's K/V rows into the caches -- a 500-event prompt costs one forward instead of 499 decode steps;
        "token" teacher-forces the prior step by step; "auto" = batched for priors of more than 32 tokens (not with
        ``return_probs``, which wants the distribution after every prior token; falls back to "token" when the prompt padded
        to a multiple of 32 rows would exceed max_seq).  The two prefill paths fill the caches through different GEMM kernels
        (same values to bf16 rounding, not bitwise), so with a fixed seed the SAMPLED continuation may differ between a
        33-token and a 32-token prompt's path: pass ``prefill`` explicitly where run-to-run identical samples matter.
        ``return_cache`` adds the per-layer (K, V) caches to the result (parity tests)."""
        st = self.store()
        st.sync_shadow()
        was_training = self.training
        self.eval()
        B, P = prior.shape
        total = P + length
        if total > self.max_seq or P < 1:
            raise ValueError(f"prior ({P}) + length ({length}) must be <= max_seq ({self.max_seq}) and prior non-empty")
        dev = st.param.device
        d, V, Vp, nl = self.embedding_dim, self.vocab_size, self.vocab_padded, self.num_layer
        bf = torch.bfloat16
        # caches are head-major [B, h, total, 64]: the decode kernel's workgroup (b, h) streams one contiguous run
        kc = [torch.zeros(B, d // 64, total, 64, dtype=bf, device=dev) for _ in range(nl)]
        vc = [torch.zeros(B, d // 64, total, 64, dtype=bf, device=dev) for _ in range(nl)]
        # ``groups`` (default 1): the batch rows as that many independent sub-batches, each with its own
        # captured step graph replayed on its own stream.  Rows never interact and the sampler draws by (seed, step, GLOBAL
        # row), so the tokens do not depend on the grouping (tests/test_gpu_decode.py).  The point would be to let one
        # sub-batch's bandwidth-bound attention stream its caches while another's chain of ~40 small projections waits out
        # its launch latencies; measured at cfg5 (profiles/README.md, round 3) two hardware queues dispatch those chains
        # SLOWER than one does (0.75 ms/token at 2 groups, 0.48 at 3-4, 0.72 at 6, against 0.50-0.51 at 1), so it stays off.
        G = max(1, min(int(groups or 1), B))
        cuts = [B * g // G for g in range(G + 1)]
        pos_all = torch.zeros(G, dtype=torch.int32, device=dev)
        tok = prior[:, 0].to(torch.int32).contiguous().to(dev)
        prior_i = prior.to(torch.int32).to(dev)
        out_tokens = torch.zeros(B, total, dtype=torch.int32, device=dev)
        out_tokens[:, :P] = prior_i
        probs_all = torch.zeros(B, total, V, dtype=torch.float32, device=dev) if return_probs else None
        probs_step = torch.zeros(B, V, dtype=torch.float32, device=dev) if return_probs else None
        hbuf = torch.empty(B, d, dtype=bf, device=dev)
        ctxbuf = torch.empty(B, d, dtype=bf, device=dev)

        class _Rows:                                        # one sub-batch: views of rows [b0, b1) of every per-row buffer
            pass
        subs = []
        for g in range(G):
            r = _Rows()
            r.b0, b1 = cuts[g], cuts[g + 1]
            r.pos = pos_all[g:g + 1]
            r.tok, r.out, r.h, r.ctx = tok[r.b0:b1], out_tokens[r.b0:b1], hbuf[r.b0:b1], ctxbuf[r.b0:b1]
            r.probs = probs_step[r.b0:b1] if return_probs else None
            r.kc, r.vc = [k[r.b0:b1] for k in kc], [v[r.b0:b1] for v in vc]
            r.ws = ops.rel_attn_decode_workspace(b1 - r.b0, total, d, dev)       # split-K partials (long caches only)
            # masked_groups (round 6 experiment): each sub-batch's stream restricted to its own 1/G of the CUs (ops.masked_stream)
            if masked_groups and G > 1:
                per = torch.cuda.get_device_properties(dev).multi_processor_count // G // 8 * 8
                r.stream = ops.masked_stream(per, g * per, dev).stream
            else:
                r.stream = torch.cuda.Stream()
            subs.append(r)
        pe = self.Decoder.pos_encoding.table()
        Pm = st.params
        layers = []
        for i in range(nl):
            pre = f"Decoder.enc_layers.{i}."
            layers.append(dict(
                wqkv=st.fused(pre + "rga.Wq.weight", pre + "rga.Wv.weight", 3 * d, d),
                bqkv=st.fused(pre + "rga.Wq.bias", pre + "rga.Wv.bias", 1, 3 * d, "param").view(3 * d),
                E=st.w(pre + "rga.E"), wfc=st.w(pre + "rga.fc.weight"), bfc=Pm[pre + "rga.fc.bias"].data,
                g1=Pm[pre + "layernorm1.weight"].data, b1=Pm[pre + "layernorm1.bias"].data,
                w1=self._layer_params()[i].wpre, bb1=self._layer_params()[i].bpre,
                w2=self._layer_params()[i].wsuf, bb2=Pm[pre + "FFN_suf.bias"].data,
                g2=Pm[pre + "layernorm2.weight"].data, b2=Pm[pre + "layernorm2.bias"].data))
        wv, bv = st.padded_view("fc.weight", Vp, d), st.padded_view("fc.bias", Vp, None, "param")
        # grammar: [V, ceil(V/32)] bit table "token v may follow token t" (e.g. REMI_EventSeq.next_token_table()); applied
        # inside the sampling kernel, so the constrained step stays graph-captured
        allow = None
        if grammar is not None:
            allow = torch.as_tensor(np.ascontiguousarray(grammar).view(np.int32) if isinstance(grammar, np.ndarray) else grammar)
            allow = allow.to(device=dev, dtype=torch.int32).contiguous()

        # decode-size batches: every LayerNorm rides in the projection that consumes it (mgx_linear_ln_fwd) and the embedding in
        # the first QKV projection (mgx_decode_embed_linear): 39 launches per token instead of 46
        fuse_ln = B <= 32 and d <= 1024
        if fuse_ln:
            # ... and every projection weight is re-laid out once per call in MFMA fragment order (ops.FragWeight): a wave load of
            # weights is then 1 KB contiguous instead of 32 B of 32 different rows
            # (the batched prefill below keeps using the row-major matrices)
            for ly in layers:
                for k in ("wqkv", "wfc", "w1", "w2"):
                    ly[k + "_f"] = ops.FragWeight(ly[k])
            wv_f = ops.FragWeight(wv)

        def step_rows(r, sample_into_out: bool):
            if fuse_ln:
                qkv, h = ops.decode_embed_linear(r.tok, Pm["Decoder.embedding.weight"].data, pe, r.pos, layers[0]["wqkv_f"],
                                                 layers[0]["bqkv"], r.h)
            else:
                h = ops.decode_embed(r.tok, Pm["Decoder.embedding.weight"].data, pe, r.pos, r.h)
                qkv = ops.linear_fwd(h, layers[0]["wqkv"], layers[0]["bqkv"], 0)
            for i, ly in enumerate(layers):
                ops.rel_attn_decode(qkv, r.kc[i], r.vc[i], ly["E"], r.pos, r.ctx, r.ws)
                nxt = layers[i + 1] if i + 1 < nl else None
                if fuse_ln:
                    a = ops.linear_fwd(r.ctx, ly["wfc_f"], ly["bfc"], 0)
                    f, o1 = ops.linear_ln_fwd(a, h, ly["g1"], ly["b1"], ly["w1_f"], ly["bb1"], 1)
                    f = ops.linear_fwd(f, ly["w2_f"], ly["bb2"], 0)
                    if nxt is not None:
                        qkv, h = ops.linear_ln_fwd(f, o1, ly["g2"], ly["b2"], nxt["wqkv_f"], nxt["bqkv"], 0)
                    else:
                        logits, h = ops.linear_ln_fwd(f, o1, ly["g2"], ly["b2"], wv_f, bv, 0)
                else:
                    a = ops.linear_fwd(r.ctx, ly["wfc"], ly["bfc"], 0)
                    o1 = ops.add_ln_fwd(a, h, ly["g1"], ly["b1"], 1e-6)[0]
                    f = ops.linear_fwd(o1, ly["w1"], ly["bb1"], 1)
                    f = ops.linear_fwd(f, ly["w2"], ly["bb2"], 0)
                    h = ops.add_ln_fwd(f, o1, ly["g2"], ly["b2"], 1e-6)[0]
                    if nxt is not None:
                        qkv = ops.linear_fwd(h, nxt["wqkv"], nxt["bqkv"], 0)
                    else:
                        logits = ops.linear_fwd(h, wv, bv, 0)
            ops.sample_topk_topp(logits, V, r.pos, r.tok, r.out if sample_into_out else None, r.probs, temperature,
                                 top_k, top_p, seed, advance=True, allow_table=allow, row0=r.b0)

        def step(sample_into_out: bool):                  # eager: the sub-batches one after the other on the current stream
            for r in subs:
                step_rows(r, sample_into_out)

        if prefill not in ("auto", "token", "batched"):
            raise ValueError("prefill must be 'auto', 'token' or 'batched'")
        if prefill == "batched" and return_probs:
            raise ValueError("return_probs needs prefill='token' (it reports the distribution after every prior token)")
        first = 0
        # the batched pass pads the prompt to a multiple of 32 rows; when that exceeds max_seq (max_seq not a multiple of 32)
        # the full-sequence kernels cannot take it: 'auto' falls back to token-by-token prefill, 'batched' says why
        fits = (P - 1 + 31) // 32 * 32 <= self.max_seq
        if prefill == "batched" and not fits:
            raise ValueError(f"prefill='batched' pads the {P - 1}-token prompt to {(P - 1 + 31) // 32 * 32} rows > max_seq={self.max_seq}")
        if prefill == "batched" or (prefill == "auto" and not return_probs and P > 32 and fits):
            # batched prefill: positions 0..P-2 through the full-sequence kernels (causal, so the zero-padded tail up to a
            # multiple of 32 cannot reach them); token P-1 then takes the ordinary decode step below
            n = P - 1
            if n > 0:
                Lp = (n + 31) // 32 * 32
                seq = torch.zeros(B, Lp, dtype=torch.int32, device=dev)
                seq[:, :n] = prior_i[:, :n]
                hh = ops.embed_pe_fwd(seq, Pm["Decoder.embedding.weight"].data, pe)
                for i, ly in enumerate(layers):
                    qkv_p = ops.linear_fwd(hh, ly["wqkv"], ly["bqkv"], 0)
                    kc[i][:, :, :n] = qkv_p[:, :n, d:2 * d].view(B, n, d // 64, 64).permute(0, 2, 1, 3)
                    vc[i][:, :, :n] = qkv_p[:, :n, 2 * d:].view(B, n, d // 64, 64).permute(0, 2, 1, 3)
                    if i + 1 == nl:
                        break                             # the last layer's output rows are not needed: token P-1 follows
                    att, _ = ops.rel_attn_fwd(qkv_p, ly["E"], None)
                    a_p = ops.linear_fwd(att, ly["wfc"], ly["bfc"], 0)
                    o1_p = ops.add_ln_fwd(a_p, hh, ly["g1"], ly["b1"], 1e-6)[0]
                    f_p = ops.linear_fwd(ops.linear_fwd(o1_p, ly["w1"], ly["bb1"], 1), ly["w2"], ly["bb2"], 0)
                    hh = ops.add_ln_fwd(f_p, o1_p, ly["g2"], ly["b2"], 1e-6)[0]
                pos_all.fill_(n)
                tok.copy_(prior_i[:, n])
                first = n
        # the (rest of the) prior is teacher-forced token by token (it also warms every kernel up before capture)
        for p in range(first, P):
            step(sample_into_out=(p == P - 1) and length > 0)
            if return_probs:
                probs_all[:, p] = probs_step
            if p + 1 < P:
                tok.copy_(prior_i[:, p + 1])
        remaining = length - 1 if length > 0 else 0
        if remaining > 0:
            if use_graph and not return_probs and remaining > 2:
                # one graph per sub-batch, each replayed on its own stream: the sub-batches never meet until the end, so their
                # chains of launches overlap freely (parallel branches INSIDE one graph were measured to run one after the other)
                cur = torch.cuda.current_stream()
                torch.cuda.synchronize()
                for r in subs:
                    r.stream.wait_stream(cur)
                    with torch.cuda.stream(r.stream):
                        r.graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(r.graph, stream=r.stream):
                            step_rows(r, True)              # the capture itself does not execute
                for p in range(remaining):
                    for r in subs:
                        with torch.cuda.stream(r.stream):
                            r.graph.replay()
                for r in subs:
                    cur.wait_stream(r.stream)
                # the graphs, their private pools and the side streams are locals: they must outlive the replays still in flight
                for r in subs:
                    r.stream.synchronize()
            else:
                for p in range(remaining):
                    step(True)
                    if return_probs:
                        probs_all[:, P + p] = probs_step
        self.train(was_training)
        res = (out_tokens, probs_all) if return_probs else out_tokens
        if return_cache:
            return (res, kc, vc)
        return res

    def test(self):
        self.eval()
        self.infer = True


def filter_probs(probs: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> torch.Tensor:
    """temperature / top-k / nucleus filtering of a probability matrix [B,V]; identity at defaults."""
    if temperature != 1.0:
        probs = torch.softmax(torch.log(probs.clamp_min(1e-30)) / temperature, -1)
    if top_k and top_k < probs.shape[-1]:
        kth = probs.topk(top_k, -1).values[:, -1:]
        probs = torch.where(probs >= kth, probs, torch.zeros_like(probs))
    if top_p < 1.0:
        sp, si = probs.sort(-1, descending=True)
        cum = sp.cumsum(-1)
        keep = (cum - sp) < top_p * cum[:, -1:]
        sp = torch.where(keep, sp, torch.zeros_like(sp))
        probs = torch.zeros_like(probs).scatter(-1, si, sp)
    return probs / probs.sum(-1, keepdim=True)
