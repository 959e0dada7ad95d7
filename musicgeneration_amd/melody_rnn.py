"""Mirror of mg/model/Event_MelodyRNN/network.py:11-164: ``Event_Melody_RNN`` (embedding -> multi-layer GRU
-> linear) with the reference's constructor, ``state_dict`` keys and ``generate`` / ``gen_forward`` /
``init_to_hidden`` / ``get_primary_event`` signatures, running each step on libmgx kernels:
row gather, two bf16 MFMA projections per layer (bias fused), one fused gate kernel per layer, the output
projection and the fused sampler; the whole step is captured in a graph when it is replayed many times.

``Train`` (teacher-forced logits) is differentiable: ``_GRUSeq`` runs the sequence's input projections as one GEMM per
layer, the recurrent projection + fused cell per step, and backward-through-time with one cell-backward kernel and one
recurrent dX GEMM per step; weight gradients are batched over the whole sequence.  Not built: packed variable-length
batches (``lengths``) and beam search (broken in the reference)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import check, ptr, stream_ptr, load as _load

BF16 = torch.bfloat16


def _pad_cols(w: torch.Tensor, k: int) -> torch.Tensor:
    out = torch.zeros(w.shape[0], k, dtype=BF16, device=w.device)
    out[:, : w.shape[1]] = w.to(BF16)
    return out.contiguous()


def _captured(ws, key, body, use_graph):
    """Run ``body()`` (a fixed sequence of kernel launches on buffers that live in ``ws``) -- through a hipGraph captured on
    first use when ``use_graph``: a layer's T time steps are T dependent launches of ~5 us each, issued from Python they
    are launch-bound (the interpreter + ctypes cost more than the kernels)."""
    if not use_graph:
        body()
        return
    g = ws["graphs"].get(key)
    if g is None:
        body()                                              # warm-up run (also the result of this call)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                body()
        torch.cuda.current_stream().wait_stream(side)
        ws["graphs"][key] = g
        return
    g.replay()


class _WsLease:
    """Marks one sequence-buffer set as holding the saved activations of a forward whose backward has not run.  Released by
    that backward, or when autograd frees the graph (the ctx drops its reference), whichever comes first."""

    def __init__(self, ws):
        self.ws = ws
        ws["busy"] = True

    def release(self):
        if self.ws is not None:
            self.ws["busy"] = False
            self.ws = None

    __del__ = release


class _GRUSeq(torch.autograd.Function):
    """Teacher-forced multi-layer GRU + output projection over a whole sequence (network.py:63-84 SeqForward with the
    primary step folded in as step 0), forward and backward-through-time on the libmgx kernels.

    forward(tokens int32 [T,B], h0 f32 [layers,B,H], pk (packed bf16 operands), ws (per-(T,B) buffers + graphs), p_drop, seed,
    *params) -> logits f32 [T,B,V].  ``params`` = embedding, (w_ih, w_hh, b_ih, b_hh) per layer, output weight, output bias:
    they are only there so that autograd routes the gradients this node returns into their ``.grad``.

    Per layer: ONE GEMM for the input projections of all time steps, then T fused step kernels (recurrent projection +
    cell, ``mgx_gru_step_fwd``) replayed from a hipGraph; backward-through-time is T fused step kernels the other way
    (``mgx_gru_step_bwd``: d_rec = dgh_{t+1} W_hh, then the cell backward), weight gradients batched over the sequence."""

    @staticmethod
    def forward(ctx, tokens, h0, pk, ws, p_drop, seed, want_grad, *params):
        T, B = tokens.shape
        nl, H = h0.shape[0], h0.shape[2]
        dev = h0.device
        tok = tokens.reshape(-1).contiguous()
        x = torch.empty(T * B, pk["Ep"], dtype=BF16, device=dev)
        check(_load().mgx_gather_rows(ptr(tok), ptr(pk["emb"]), ptr(x), T * B, pk["Ep"], pk["emb"].shape[0], stream_ptr()),
              "mgx_gather_rows")
        # a grad-enabled forward leases the buffer set until its backward (Train hands out a set that is not leased)
        lease = _WsLease(ws) if want_grad else None       # (grad mode is always off INSIDE a Function.forward)
        xs = []
        for l, ly in enumerate(pk["layers"]):
            wl = ws["layers"][l]
            gi, gh, h_all, hp = wl["gi"], wl["gh"], wl["h_all"], wl["hp"]
            gi.view(T * B, 3 * H).copy_(ops.linear_fwd(x, ly["wih"], ly["bih"], 0))     # all time steps in one GEMM
            h_all[0].copy_(h0[l])
            hp[0].copy_(h0[l])                               # hp[t] = bf16(h_{t-1}); y = hp[1:]

            def steps(gi=gi, gh=gh, h_all=h_all, hp=hp, ly=ly):
                for t in range(T):
                    ops.gru_step_fwd(gi[t], hp[t], h_all[t], ly["whh_f"], ly["bhh"], h_all[t + 1], hp[t + 1], gh[t])
            _captured(ws, ("fwd", l), steps, ws["use_graph"])
            xs.append(x)
            x = hp[1:].reshape(T * B, H)
            if l < nl - 1:
                x = ops.dropout_bf16(x, p_drop, seed + l)
        logits = ops.linear_fwd(x, pk["wo"], pk["bo"], 0)
        ctx.pk, ctx.ws, ctx.lease, ctx.xs = pk, ws, lease, xs
        ctx.cfg, ctx.x_last, ctx.tok = (T, B, H, nl, p_drop, seed), x, tok
        ctx.shapes = [p.shape for p in params]
        V = params[0].shape[0]
        return logits[:, :V].float().view(T, B, V)

    @staticmethod
    def backward(ctx, dlogits):
        pk, ws = ctx.pk, ctx.ws
        if ctx.lease is None or ctx.lease.ws is not ws:
            raise RuntimeError("Event_Melody_RNN.Train: backward through a forward that ran without gradients enabled, or twice "
                               "through the same forward (its sequence buffers have been handed back)")
        T, B, H, nl, p_drop, seed = ctx.cfg
        dev = dlogits.device
        V, Vp = ctx.shapes[0][0], pk["wo"].shape[0]
        dl = torch.zeros(T * B, Vp, dtype=BF16, device=dev)
        dl[:, :V] = dlogits.reshape(T * B, V)
        g_wo = torch.zeros(Vp, H, device=dev)
        g_bo = torch.zeros(Vp, device=dev)
        ops.linear_dw(dl, ctx.x_last, g_wo, g_bo)
        dx = ops.linear_dx(dl, pk["wo"])                                            # [T*B, H]
        dh0 = torch.empty(nl, B, H, device=dev)
        layer_grads = [None] * nl
        for l in reversed(range(nl)):
            ly, wl = pk["layers"][l], ws["layers"][l]
            gi, gh, h_all, hp = wl["gi"], wl["gh"], wl["h_all"], wl["hp"]
            dgi, dgh, dy, dh_dir, dh0_l = wl["dgi"], wl["dgh"], wl["dy"], wl["dh"], wl["dh0"]
            if l < nl - 1:
                dx = ops.dropout_bf16(dx, p_drop, seed + l)
            dy.view(T * B, H).copy_(dx)

            def steps(gi=gi, gh=gh, h_all=h_all, dgi=dgi, dgh=dgh, dy=dy, dh_dir=dh_dir, dh0_l=dh0_l, ly=ly):
                for t in reversed(range(T)):
                    last = t == T - 1
                    ops.gru_step_bwd(gi[t], gh[t], h_all[t], None if last else dh_dir[(t + 1) & 1], None if last else dgh[t + 1],
                                     ly["whhT_f"], dy[t], dgi[t], dgh[t], dh_dir[t & 1])
                ops.gru_step_bwd(None, None, None, dh_dir[0], dgh[0], ly["whhT_f"], None, None, None, dh0_l, final=True)
            _captured(ws, ("bwd", l), steps, ws["use_graph"])
            dh0[l] = dh0_l
            in_p = ly["wih"].shape[1]
            g_wih = torch.zeros(3 * H, in_p, device=dev)
            g_whh = torch.zeros(3 * H, H, device=dev)
            g_bih = torch.zeros(3 * H, device=dev)
            g_bhh = torch.zeros(3 * H, device=dev)
            ops.linear_dw_grouped([(dgi.view(T * B, 3 * H), ctx.xs[l], g_wih, g_bih),
                                   (dgh.view(T * B, 3 * H), hp[:-1].reshape(T * B, H), g_whh, g_bhh)])
            layer_grads[l] = (g_wih, g_whh, g_bih, g_bhh)
            dx = ops.linear_dx(dgi.view(T * B, 3 * H), ly["wih"])                   # gradient of the layer's input
        g_emb = torch.zeros(ctx.shapes[0], device=dev)
        ops.scatter_add_rows(ctx.tok, dx, g_emb)
        grads = [g_emb]
        for l in range(nl):
            g_wih, g_whh, g_bih, g_bhh = layer_grads[l]
            grads += [g_wih[:, : ctx.shapes[1 + 4 * l][1]], g_whh, g_bih, g_bhh]
        grads += [g_wo[:V], g_bo[:V]]
        ctx.xs = None
        ctx.lease.release()
        return (None, dh0, None, None, None, None, None) + tuple(grads)


class _GruState:
    """hidden state of the sampling step: two (f32, bf16) buffer pairs used alternately -- the fused layer step
    (mgx_gru_step_x_fwd) cannot update h in place, other workgroups still read h_{t-1} as their operand."""

    def __init__(self, h32):
        self.h32 = [h32.detach().float().contiguous().clone(), torch.empty_like(h32, dtype=torch.float32)]
        self.hbf = [self.h32[0].to(BF16), torch.empty_like(h32, dtype=BF16)]
        self.cur = 0

    @property
    def h(self):
        return self.h32[self.cur]


class Event_Melody_RNN(nn.Module):
    def __init__(self, init_dim, event_dim, hidden_dim, rnn_layers=2, dropout=0.5):
        super().__init__()
        self.event_dim = event_dim
        self.init_dim = init_dim
        self.hidden_dim = hidden_dim
        self.rnn_layers = rnn_layers
        self.output_dim = event_dim
        self.primary_event = self.event_dim - 1
        self.inithid_fc = nn.Linear(init_dim, rnn_layers * hidden_dim)
        self.inithid_fc_activation = nn.Tanh()
        self.event_embedding = nn.Embedding(event_dim, event_dim)
        self.rnn = nn.GRU(self.event_dim, self.hidden_dim, num_layers=rnn_layers, dropout=dropout)
        self.output_fc = nn.Linear(hidden_dim, self.output_dim)
        self.output_fc_activation = nn.Softmax(dim=-1)
        if hidden_dim % 64:
            raise ValueError("hidden_dim must be a multiple of 64 for the MFMA projections")
        self._packed = None
        self._train_ws = {}

    # ---- bf16 operand pack (refreshed in place when parameters change) -----------------------------------------
    def _pack(self):
        """bf16 operand copies of the parameters.  The buffers are allocated ONCE per device and refreshed in place when a
        parameter changes, so that hipGraphs that captured their addresses (generate's step, Train's time loops) stay valid
        across optimiser steps."""
        ver = tuple(p._version for p in self.parameters())
        dev = self.output_fc.weight.device
        pk = self._packed
        if pk is not None and pk["ver"] == ver and pk["dev"] == dev:
            return pk
        if dev.type != "cuda":
            raise ops._lib.MgxError("Event_Melody_RNN runs on the MI355X kernels only: move it to a HIP device")
        H, nl = self.hidden_dim, self.rnn_layers
        Ep = (self.event_dim + 63) // 64 * 64            # embedding width padded to the GEMM's K % 64
        Vp = (self.event_dim + 7) // 8 * 8               # output rows padded to N % 8 (forward needs 4, dX/dW 8)
        if pk is None or pk["dev"] != dev:
            pk = {"dev": dev, "Ep": Ep, "Vp": Vp, "emb": torch.zeros(self.event_dim, Ep, dtype=BF16, device=dev), "layers": [],
                  "wo": torch.zeros(Vp, H, dtype=BF16, device=dev), "bo": torch.zeros(Vp, dtype=torch.float32, device=dev)}
            for l in range(nl):
                in_p = Ep if l == 0 else H
                pk["layers"].append(dict(wih=torch.zeros(3 * H, in_p, dtype=BF16, device=dev),
                                         whh=torch.empty(3 * H, H, dtype=BF16, device=dev),
                                         # fragment-ordered copies for the fused step kernels (ops.pack_frag)
                                         wih_f=torch.zeros(3 * H, in_p, dtype=BF16, device=dev),
                                         whh_f=torch.empty(3 * H, H, dtype=BF16, device=dev),
                                         whhT_f=torch.empty(H, 3 * H, dtype=BF16, device=dev),
                                         bih=torch.empty(3 * H, dtype=torch.float32, device=dev),
                                         bhh=torch.empty(3 * H, dtype=torch.float32, device=dev)))
            self._train_ws = {}
        with torch.no_grad():
            pk["emb"][:, : self.event_dim].copy_(self.event_embedding.weight.data)
            for l, ly in enumerate(pk["layers"]):
                wih = getattr(self.rnn, f"weight_ih_l{l}").data
                whh = getattr(self.rnn, f"weight_hh_l{l}").data
                ly["wih"][:, : wih.shape[1]].copy_(wih)
                ly["whh"].copy_(whh)
                ly["wih_f"].copy_(ops.pack_frag(ly["wih"]))
                ly["whh_f"].copy_(ops.pack_frag(whh))
                ly["whhT_f"].copy_(ops.pack_frag(whh.t()))
                ly["bih"].copy_(getattr(self.rnn, f"bias_ih_l{l}").data)
                ly["bhh"].copy_(getattr(self.rnn, f"bias_hh_l{l}").data)
            pk["wo"][: self.event_dim].copy_(self.output_fc.weight.data)
            pk["bo"][: self.event_dim].copy_(self.output_fc.bias.data)
        pk["ver"] = ver
        self._packed = pk
        return pk

    def _train_workspace(self, pk, T, B):
        """Sequence buffers of Train for one (T, B) -- they cross the boundary of the captured time loops, so they must keep
        their addresses between calls -- and the hipGraphs of those loops.  A buffer set holds the saved activations of a
        forward until its backward has run (``_WsLease``), so a second forward of the same shape before that backward --
        ``(loss1 + loss2).backward()``, an evaluation between forward and backward -- gets ANOTHER set: up to three per shape
        are kept (each with its own graphs); beyond that a throw-away set with eager launches is used.  Calls under
        ``no_grad`` take any free set and lease nothing.  At most eight shapes are kept."""
        import os
        cache = self._train_ws
        sets = cache.get((T, B))
        if sets is None:
            if len(cache) >= 8:
                cache.pop(next(iter(cache)))
            sets = cache[(T, B)] = []
        for ws in sets:
            if not ws["busy"]:
                return ws
        dev, H = pk["dev"], self.hidden_dim
        mk = lambda *shape, dt=BF16: torch.empty(*shape, dtype=dt, device=dev)
        keep = len(sets) < 3
        ws = {"busy": False, "graphs": {}, "use_graph": keep and os.environ.get("MGX_GRU_GRAPH", "1") != "0", "layers": [
            dict(gi=mk(T, B, 3 * H), gh=mk(T, B, 3 * H), h_all=mk(T + 1, B, H, dt=torch.float32), hp=mk(T + 1, B, H),
                 dgi=mk(T, B, 3 * H), dgh=mk(T, B, 3 * H), dy=mk(T, B, H), dh=mk(2, B, H, dt=torch.float32),
                 dh0=mk(B, H, dt=torch.float32)) for _ in range(self.rnn_layers)]}
        if keep:
            sets.append(ws)
        return ws

    # ---- reference API ------------------------------------------------------------------------------
    def get_primary_event(self, batch_size):
        return torch.full((1, batch_size), self.primary_event, dtype=torch.long, device=self.output_fc.weight.device)

    def init_to_hidden(self, init):
        """network.py:98-104 (one tiny projection, once per sequence; library op)"""
        batch_size = init.shape[0]
        out = self.inithid_fc_activation(self.inithid_fc(init))
        return out.view(self.rnn_layers, batch_size, self.hidden_dim)

    @torch.no_grad()
    def _step(self, pk, tok_i32, st, xbuf):
        """event int32 [B] -> logits bf16 [B,Vp]; ``st`` (_GruState, [layers,B,H]) advances one step: per layer ONE launch
        (both projections + the cell, network.py:144-149), reading st's current buffers and writing the other pair."""
        lib = _load()
        B = tok_i32.numel()
        check(lib.mgx_gather_rows(ptr(tok_i32), ptr(pk["emb"]), ptr(xbuf), B, pk["Ep"], self.event_dim, stream_ptr()),
              "mgx_gather_rows")
        x = xbuf
        c, n = st.cur, st.cur ^ 1
        for l, ly in enumerate(pk["layers"]):
            ops.gru_step_x_fwd(x, ly["wih_f"], ly["bih"], st.hbf[c][l], st.h32[c][l], ly["whh_f"], ly["bhh"], st.h32[n][l], st.hbf[n][l])
            x = st.hbf[n][l]
        st.cur = n
        return ops.linear_fwd(x, pk["wo"], pk["bo"], 0)

    @torch.no_grad()
    def gen_forward(self, event, hidden=None):
        """One step (network.py:51-61): event int64 [1,B], hidden [layers,B,H] -> (logits [1,B,V] f32, hidden')"""
        assert len(event.shape) == 2 and event.shape[0] == 1
        pk = self._pack()
        B = event.shape[1]
        dev = pk["dev"]
        st = _GruState(torch.zeros(self.rnn_layers, B, self.hidden_dim, device=dev) if hidden is None else hidden.to(dev))
        xbuf = torch.empty(B, pk["Ep"], dtype=BF16, device=dev)
        logits = self._step(pk, event[0].to(torch.int32).contiguous(), st, xbuf)
        return logits[:, : self.event_dim].float().unsqueeze(0), st.h

    def forward(self, event, hidden=None):
        return self.gen_forward(event, hidden)

    def Train(self, init, events, lengths=None):
        """Teacher-forced logits [T+1,B,V] (network.py:63-84,109-116): the primary event, then ``events``, through the
        GRU from ``init_to_hidden(init)``; differentiable (backward-through-time on the libmgx kernels), with nn.GRU's
        inter-layer dropout in training mode.

        ``lengths`` (the reference's ``sequence`` mode, train.py:263-287): ``events`` is then SeqBatchify's BATCH-FIRST
        ``X [B,Tmax]`` (rows sorted by length, zero padded) and the result is batch-first ``[B, Tmax+1, V]``: the
        primary-event step followed by one step per event, where steps past a row's length hold ``output_fc(0)`` --
        exactly what ``pack_padded_sequence`` -> GRU -> ``pad_packed_sequence`` -> ``output_fc`` gives (a GRU is causal
        per batch row, so running the padded rows and overwriting the steps past the end is the packed computation;
        the padded steps send no gradient into the recurrence).  The reference's own call mixes the batch and time axes
        (network.py:64,71 take ``events.shape[1]`` as the batch size of a batch-first tensor) and cannot run; this is the
        computation its loss (``flatten_padded_sequences`` + concatenated labels, utils/data.py:14-36) is written for."""
        if lengths is not None:
            ev = torch.as_tensor(events)
            if ev.dim() != 2 or ev.shape[0] != init.shape[0] or len(lengths) != ev.shape[0]:
                raise ValueError("with lengths, events must be batch-first [B,Tmax] (SeqBatchify's X) and len(lengths) == B")
            lens = torch.as_tensor(lengths, dtype=torch.int64)
            if int(lens.max()) > ev.shape[1] or int(lens.min()) < 1:
                raise ValueError("lengths must lie in [1, Tmax]")
            # Tmax is rounded up to a multiple of 16 steps (zero events, computed and dropped): the time loops are replayed from
            # hipGraphs keyed by (T, B), and a data set of ragged batches would otherwise capture a new graph per distinct Tmax
            Tmax = ev.shape[1]
            Tb = (Tmax + 15) // 16 * 16
            evt = ev.t().contiguous()
            if Tb != Tmax:
                evt = torch.cat([evt, torch.zeros(Tb - Tmax, evt.shape[1], dtype=evt.dtype, device=evt.device)], 0)
            full = self.Train(init, evt)[: Tmax + 1]                         # [Tmax+1, B, V], padded steps included
            steps = torch.arange(full.shape[0], device=full.device)[:, None]
            valid = steps <= lens.to(full.device)[None, :]                   # step 0 = primary event, steps 1..len = events
            fill = self.output_fc.bias.to(full.dtype)                        # output_fc applied to a zero (padded) GRU output
            return torch.where(valid[..., None], full, fill.expand_as(full)).transpose(0, 1)
        pk = self._pack()
        hidden = self.init_to_hidden(init).float().contiguous()
        B = init.shape[0]
        tokens = torch.cat([self.get_primary_event(B), events.to(pk["dev"]).long()], 0).to(torch.int32).contiguous()
        p_drop = float(self.rnn.dropout) if self.training else 0.0
        self._train_calls = getattr(self, "_train_calls", 0) + 1
        params = [self.event_embedding.weight]
        for l in range(self.rnn_layers):
            params += [getattr(self.rnn, f"{n}_l{l}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        params += [self.output_fc.weight, self.output_fc.bias]
        seed = (torch.initial_seed() + 7919 * self._train_calls) & 0x7FFFFFFFFFFF
        ws = self._train_workspace(pk, tokens.shape[0], B)
        return _GRUSeq.apply(tokens, hidden, pk, ws, p_drop, seed, torch.is_grad_enabled(), *params)

    @torch.no_grad()
    def generate(self, init, steps, events=None, greedy=1.0, temperature=1.0, teacher_forcing_ratio=1.0,
                 output_type='index', verbose=False, seed=0, use_graph=True):
        """network.py:119-164.  ``greedy`` is the probability of an arg-max step (host coin per step, as in the
        reference); otherwise the event is drawn from softmax(logits / temperature) by the fused sampler."""
        batch_size = init.shape[0]
        assert init.shape[1] == self.init_dim and steps > 0
        use_teacher_forcing = events is not None
        if use_teacher_forcing:
            assert len(events.shape) == 2 and events.shape[0] >= steps - 1
            events = events[:steps - 1]
        pk = self._pack()
        dev, V = pk["dev"], self.event_dim
        st = _GruState(self.init_to_hidden(init))
        xbuf = torch.empty(batch_size, pk["Ep"], dtype=BF16, device=dev)
        tok = torch.full((batch_size,), self.primary_event, dtype=torch.int32, device=dev)
        pos = torch.zeros(1, dtype=torch.int32, device=dev)
        probs = torch.empty(batch_size, V, device=dev) if output_type == 'softmax' else None
        rng = np.random.RandomState(seed)
        coins = rng.random_sample(steps) < greedy
        tf_coins = rng.random_sample(steps) <= teacher_forcing_ratio
        outputs = []
        plain = output_type == 'index' and not use_teacher_forcing
        out_tokens = torch.zeros(batch_size, steps + 1, dtype=torch.int32, device=dev) if plain else None

        def one(step_greedy):
            logits = self._step(pk, tok, st, xbuf)
            ops.sample_topk_topp(logits, V, pos, tok, out_tokens, probs, temperature, 1 if step_greedy else 0, 1.0, seed,
                                 advance=True)
            return logits

        graphs = {}
        for step in range(steps):
            g = bool(coins[step])
            if plain and use_graph and steps > 8 and step >= 2:
                key = (g, st.cur)                           # a graph reads one state-buffer pair and writes the other
                if key not in graphs:
                    torch.cuda.synchronize()
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        gr = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gr, stream=side):
                            one(g)
                    torch.cuda.current_stream().wait_stream(side)
                    st.cur ^= 1                             # the capture only recorded the step
                    graphs[key] = gr
                graphs[key].replay()
                st.cur ^= 1
                continue
            logits = one(g)
            if output_type == 'index':
                outputs.append(tok.clone().long().unsqueeze(0))
            elif output_type == 'softmax':
                outputs.append(probs.clone().unsqueeze(0))
            elif output_type == 'logit':
                outputs.append(logits[:, :V].float().unsqueeze(0))
            else:
                assert False
            if use_teacher_forcing and step < steps - 1 and tf_coins[step]:
                tok.copy_(events[step].to(torch.int32))
        if plain:
            return out_tokens[:, 1:].t().contiguous().long()
        return torch.cat(outputs, 0)

    @torch.no_grad()
    def beam_search(self, init, steps, beam_size, temperature=1.0, stochastic=False, verbose=False, seed=0):
        """network.py:168-268, repaired.  The reference's version cannot run (it gathers the hidden state with a
        hard-coded ``repeat(4, 1, 1, 1024)``) and scores beams with raw logits; this one keeps its interface and
        its structure -- every step advances batch*beam rows through the GRU step kernels, expands each beam by all events
        and keeps the ``beam_size`` best -- but scores with log-softmax(logits / temperature), starts from ONE live beam (the
        others at -inf, so duplicates of the first expansion cannot fill the beam) and re-orders the hidden state by the
        surviving parents.  ``stochastic=True`` selects survivors by Gumbel-perturbed scores (the reference's variant, with
        the true scores carried along).  Returns the best sequence per batch row, int64 [steps, batch]."""
        assert len(init.shape) == 2 and init.shape[1] == self.init_dim
        assert self.event_dim >= beam_size > 0 and steps > 0
        pk = self._pack()
        dev, V, nl, H = pk["dev"], self.event_dim, self.rnn_layers, self.hidden_dim
        B, K = init.shape[0], beam_size
        h32 = self.init_to_hidden(init).detach().float()                          # [layers, B, H]
        st = _GruState(h32[:, :, None, :].repeat(1, 1, K, 1).reshape(nl, B * K, H))
        xbuf = torch.empty(B * K, pk["Ep"], dtype=BF16, device=dev)
        tok = torch.full((B * K,), self.primary_event, dtype=torch.int32, device=dev)
        score = torch.full((B, K), float("-inf"), device=dev)
        score[:, 0] = 0.0
        seqs = torch.zeros(B, K, 0, dtype=torch.long, device=dev)
        gen = torch.Generator(device=dev).manual_seed(seed)
        rows = torch.arange(B, device=dev)[:, None]
        for _ in range(steps):
            logits = self._step(pk, tok, st, xbuf)[:, :V].float()                 # [B*K, V]
            logp = torch.log_softmax(logits / temperature, -1).view(B, K, V)
            cand = (score[:, :, None] + logp).view(B, K * V)                      # all expansions of all live beams
            if stochastic:
                u = torch.rand(cand.shape, device=dev, generator=gen).clamp_(1e-20, 1.0)
                pick = (cand - torch.log(-torch.log(u))).topk(K, -1).indices
            else:
                pick = cand.topk(K, -1).indices
            score = cand.gather(-1, pick)
            parent, event = pick // V, pick % V                                   # [B, K]
            seqs = torch.cat([seqs[rows, parent], event[:, :, None]], -1)
            flat = (rows * K + parent).reshape(-1)
            st = _GruState(st.h[:, flat])                                         # surviving parents' states
            tok = event.reshape(-1).to(torch.int32).contiguous()
        best = seqs[torch.arange(B, device=dev), score.argmax(-1)]                # [B, steps]
        return best.t().contiguous()

