#!/usr/bin/env python3
"""Generate golden fixtures from the reference (runs ONLY in the build container).

The reference at /root/reference is imported read-only (PYTHONDONTWRITEBYTECODE=1), with
inert stubs for third-party modules that are absent from this image (pretty_midi,
miditoolkit, torchvision, tensorboardX, progress).  Only inputs and outputs are stored;
no reference source text is copied.  Two invocations are needed because the reference has
two different top-level modules called ``utils``:

    python tests/golden/gen_golden.py mt      # MusicTransformer model/loss/schedule (G1-G4,G6,G7,G9)
    python tests/golden/gen_golden.py codec   # codecs + Event_Melody_RNN (G5, G8)

Fixtures are small .npz / .json files in tests/golden/.
"""
import json
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mg/model"


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    mod("pretty_midi", PrettyMIDI=_Dummy, Note=_Dummy, Instrument=_Dummy)
    mt = mod("miditoolkit")
    mt.midi = mod("miditoolkit.midi")
    mod("torchvision")
    mod("tensorboardX", SummaryWriter=_Dummy)

    class Bar:
        def __init__(self, *a, **k):
            pass

        def iter(self, it):
            return it

    p = mod("progress")
    p.bar = mod("progress.bar", Bar=Bar)


def gen_mt():
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(REF, "MusicTransformer"))
    import config  # noqa
    import layers  # noqa
    import network  # noqa
    import criterion  # noqa
    import metrics  # noqa
    import utils  # noqa

    torch.set_num_threads(4)
    out = {}

    def npy(t):
        return t.detach().cpu().numpy()

    # ---- G1: RelativeGlobalAttention unit -------------------------------------------------
    for tag, (B, h, L, M, dh) in {"a": (2, 2, 16, 16, 8), "b": (2, 2, 12, 16, 8)}.items():
        torch.manual_seed(0)
        d = h * dh
        rga = layers.RelativeGlobalAttention(h=h, d=d, max_seq=M)
        x = torch.randn(B, L, d, requires_grad=True)
        tok = torch.randint(0, 5, (B, L))
        tok[0, -2:] = 4  # pretend 4 is pad: key-padding columns
        mask = (tok == 4)[:, None, None, :] | ~torch.tril(torch.ones(L, L, dtype=torch.bool))
        o, w = rga([x, x, x], mask)
        go = torch.randn_like(o)
        (o * go).sum().backward()
        g = {"x": npy(x), "mask": npy(mask), "out": npy(o), "w": npy(w), "go": npy(go),
             "gx": npy(x.grad), "gE": npy(rga.E.grad), "gWq": npy(rga.Wq.weight.grad),
             "gWk": npy(rga.Wk.weight.grad), "gWv": npy(rga.Wv.weight.grad)}
        for k, v in rga.state_dict().items():
            g["p." + k] = npy(v)
        np.savez_compressed(os.path.join(HERE, f"g1{tag}_rga.npz"), **g)

    # ---- G2: tiny full model -------------------------------------------------------------
    torch.manual_seed(0)
    V, d, nl, L = config.vocab_size, 128, 2, 32
    pad = config.pad_token
    mt = network.MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
    gen = torch.Generator().manual_seed(1)
    xfull = torch.randint(0, V - 1, (3, L + 1), generator=gen)
    xfull[1, -4:] = pad   # trailing pads
    xfull[2, -1:] = pad
    x = xfull[:, :-1].to(torch.int)
    y = xfull[:, 1:].to(torch.int)
    mt.train()
    logits = mt(x)
    lossf = criterion.SmoothCrossEntropyLoss(config.label_smooth, V, pad)
    ms = metrics.MetricsSet({"accuracy": metrics.CategoricalAccuracy(), "loss": lossf,
                             "bucket": metrics.LogitsBucketting(V)})
    m = ms(logits, y)
    m["loss"].backward()
    g = {"x": npy(x), "y": npy(y), "logits": npy(logits), "loss": npy(m["loss"]),
         "accuracy": npy(m["accuracy"]), "bucket": npy(m["bucket"])}
    for k, v in mt.state_dict().items():
        g["p." + k] = npy(v)
    for k, p in mt.named_parameters():
        g["g." + k] = npy(p.grad)
    mt.eval()
    with torch.no_grad():
        elog, ws = mt(x)
    g["eval_logits"] = npy(elog)
    g["eval_w0"] = npy(ws[0])
    g["eval_w1"] = npy(ws[1])
    # G7: sampler distributions: reference generate() semantics (mask=None) and causal last row
    with torch.no_grad():
        prior = x[:2, :9].long()
        res, _ = mt.Decoder(prior, None)
        g["g7_prior"] = npy(prior)
        g["g7_nomask_probs"] = npy(mt.fc(res).softmax(-1)[:, -1])
        g["g7_causal_probs"] = npy(elog.softmax(-1))  # [B,L,V] causal probabilities for every prefix
    np.savez_compressed(os.path.join(HERE, "g2_model.npz"), **g)

    # ---- G2b: leading-pad row (fully masked rows -> uniform attention) ---------------------
    torch.manual_seed(0)
    mt2 = network.MusicTransformer(embedding_dim=64, vocab_size=V, num_layer=1, max_seq=16, dropout=0.0)
    xb = torch.randint(0, V - 1, (2, 16), generator=gen).to(torch.int)
    xb[0, :3] = pad
    mt2.train()
    lg = mt2(xb)
    gb = {"x": npy(xb), "logits": npy(lg)}
    for k, v in mt2.state_dict().items():
        gb["p." + k] = npy(v)
    np.savez_compressed(os.path.join(HERE, "g2b_leadpad.npz"), **gb)

    # ---- G3: SmoothCrossEntropyLoss -------------------------------------------------------
    torch.manual_seed(3)
    lg = torch.randn(3, 7, V, requires_grad=True) * 3
    lg.retain_grad()
    tg = torch.randint(0, V - 1, (3, 7))
    tg[0, 5:] = pad
    tg[2, 6] = pad
    ls = lossf(lg, tg)
    ls.backward()
    np.savez_compressed(os.path.join(HERE, "g3_smoothce.npz"), logits=npy(lg), target=npy(tg),
                        loss=npy(ls), glogits=npy(lg.grad), eps=np.float32(config.label_smooth),
                        pad=np.int64(pad))

    # ---- G4: CustomSchedule ---------------------------------------------------------------
    steps = [1, 2, 100, 3999, 4000, 4001, 8000]
    out["g4"] = {str(dm): [criterion.CustomSchedule(dm).rate(s) for s in steps] for dm in (256, 512)}
    out["g4_steps"] = steps

    # ---- G6: mask + positional table ------------------------------------------------------
    xm = torch.tensor([[1, 2, pad, 4, 5, pad, pad, 3], [pad, 1, 2, 3, 4, 5, 6, 7]])
    _, _, lam = utils.get_masked_with_pad_tensor(8, xm, xm, pad)
    pe = layers.DynamicPositionEmbedding(16, max_seq=8).positional_embedding
    np.savez_compressed(os.path.join(HERE, "g6_mask_pe.npz"), x=npy(xm), mask=npy(lam), pe=pe,
                        pad=np.int64(pad))

    # ---- G9: 3 optimizer steps with accum_grad=2 ------------------------------------------
    torch.manual_seed(0)
    mt3 = network.MusicTransformer(embedding_dim=64, vocab_size=V, num_layer=2, max_seq=16, dropout=0.0)
    opt = torch.optim.Adam(mt3.parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9)
    sch = criterion.CustomSchedule(64, optimizer=opt)
    g9 = {}
    for k, v in mt3.state_dict().items():
        g9["p0." + k] = npy(v).copy()
    gen9 = torch.Generator().manual_seed(9)
    xs, losses, lrs = [], [], []
    opt.zero_grad()
    mt3.train()
    for it in range(6):
        xf = torch.randint(0, V - 1, (2, 17), generator=gen9)
        xs.append(npy(xf))
        lg = mt3(xf[:, :-1].to(torch.int))
        loss = lossf(lg, xf[:, 1:].to(torch.int)) / 2
        loss.backward()
        losses.append(float(loss) * 2)
        if (it + 1) % 2 == 0:
            sch.step()
            lrs.append(sch._rate)
            opt.zero_grad()
    g9["xs"] = np.stack(xs)
    g9["losses"] = np.array(losses, dtype=np.float64)
    g9["lrs"] = np.array(lrs, dtype=np.float64)
    for k, v in mt3.state_dict().items():
        g9["p3." + k] = npy(v)
    np.savez_compressed(os.path.join(HERE, "g9_optim.npz"), **g9)

    with open(os.path.join(HERE, "g4_schedule.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("mt fixtures written")


def gen_codec():
    import numpy as np
    import torch

    sys.path.insert(0, REF)
    # utils/REMI.py and MuMIDI.py import utils.chord_inference (needs miditoolkit stub only)
    from utils.sequence import EventSeq, Event as MEvent
    from utils import REMI, MuMIDI

    out = {}
    # MIDI-like
    fr = EventSeq.feat_ranges()
    out["midi_like"] = {
        "dim": EventSeq.dim(),
        "feat_dims": list(EventSeq.feat_dims().items()),
        "feat_ranges": [(k, r.start, r.stop) for k, r in fr.items()],
    }
    ids = np.array([0, 87, 88, 175, 176, 207, 208, 307, 250, 30, 118, 209, 209, 190], dtype=np.uint16)
    es = EventSeq.from_array(ids)
    out["midi_like"]["from_array_ids"] = ids.tolist()
    out["midi_like"]["from_array_events"] = [(e.type, float(e.time), int(e.value)) for e in es.events]
    out["midi_like"]["to_array"] = es.to_array().tolist()
    out["midi_like"]["to_array_dtype"] = str(es.to_array().dtype)
    allids = np.arange(EventSeq.dim(), dtype=np.uint16)
    out["midi_like"]["roundtrip_all"] = bool((EventSeq.from_array(allids).to_array() == allids).all())
    out["midi_like"]["velocity_bins"] = EventSeq.get_velocity_bins().tolist()
    out["midi_like"]["time_shift_bins"] = EventSeq.time_shift_bins.tolist()

    # REMI
    R = REMI.REMI_EventSeq
    out["remi"] = {
        "dim": R.dim(),
        "feat_dims": list(R.feat_dims().items()),
        "feat_ranges": [(k, r.start, r.stop) for k, r in R.feat_ranges().items()],
        "table": [(e.name, e.value) for e in R.from_array(np.arange(R.dim()))],
        "roundtrip_all": bool((R.to_array(R.from_array(np.arange(R.dim()))) == np.arange(R.dim())).all()),
        "to_array_dtype": str(R.to_array(R.from_array(np.arange(4))).dtype),
        "chord_map": list(REMI.chord_map.items()),
    }
    script = [("bar", 0), ("position", 0), ("tempo_class", 1), ("tempo_value", 30), ("chord", "C:maj"),
              ("position", 4), ("note_velocity", 3), ("note_on", 60), ("note_duration", 7),
              ("chord", "N:N"), ("position", 15), ("note_velocity", 0), ("note_on", 126), ("note_duration", 63)]
    out["remi"]["script"] = script
    out["remi"]["script_ids"] = R.to_array([REMI.Event(n, None, v, None) for n, v in script]).tolist()
    try:
        R.to_array([REMI.Event("note_velocity", None, 4, None)])
        out["remi"]["velocity4"] = "ok"
    except Exception as e:  # quirk (i)
        out["remi"]["velocity4"] = type(e).__name__

    # MuMIDI
    Mu = MuMIDI.MuMIDI_EventSeq
    out["mumidi"] = {
        "dim": Mu.dim(),
        "feat_dims": list(Mu.feat_dims().items()),
        "feat_ranges": [(k, r.start, r.stop) for k, r in Mu.feat_ranges().items()],
        "table": [(e.name, e.value) for e in Mu.from_array(np.arange(Mu.dim()))],
        "to_array_dtype": str(Mu.to_array(Mu.from_array(np.arange(4))).dtype),
        "track_ids": {t: int(Mu.get_track_id(t)) for t in MuMIDI.DEFAULT_TRACKS},
    }
    script = [("bar", 0), ("position", 0), ("tempo_class", 2), ("tempo_value", 59), ("chord", "A#:dom"),
              ("track_melody", 0), ("position", 32), ("note_on", 255), ("note_duration", 31),
              ("note_velocity", 31), ("track_drum", 5), ("empty", 0), ("note_on", 0)]
    out["mumidi"]["script"] = script
    out["mumidi"]["script_ids"] = Mu.to_array([MuMIDI.Event(n, None, v, None) for n, v in script]).tolist()
    try:
        Mu.to_array(Mu.from_array(np.arange(Mu.dim())))
        out["mumidi"]["roundtrip_all"] = "ok"
    except Exception as e:  # quirk (ii)
        out["mumidi"]["roundtrip_all"] = type(e).__name__
    nt = np.array([i for i in range(Mu.dim()) if not (355 <= i <= 360)])
    out["mumidi"]["roundtrip_nontrack"] = bool((Mu.to_array(Mu.from_array(nt)) == nt).all())

    with open(os.path.join(HERE, "g5_codecs.json"), "w") as f:
        json.dump(out, f)

    # ---- G8: Event_Melody_RNN -------------------------------------------------------------
    from Event_MelodyRNN.network import Event_Melody_RNN

    torch.manual_seed(0)
    net = Event_Melody_RNN(init_dim=8, event_dim=40, hidden_dim=64, rnn_layers=2, dropout=0.0)
    net.eval()
    init = torch.randn(3, 8)
    g = {"init": init.numpy()}
    with torch.no_grad():
        hid = net.init_to_hidden(init)
        g["hid0"] = hid.numpy()
        ev = net.get_primary_event(3)
        evs = [torch.tensor([[5, 6, 7]]), torch.tensor([[1, 0, 39]])]
        for s in range(3):
            o, hid = net.gen_forward(ev, hid)
            g[f"step{s}_event"] = ev.numpy()
            g[f"step{s}_logits"] = o.numpy()
            g[f"step{s}_hidden"] = hid.numpy()
            if s < 2:
                ev = evs[s]
        events = torch.randint(0, 40, (6, 3))
        g["train_events"] = events.numpy()
        g["train_logits"] = net.Train(init, events).numpy()
    for k, v in net.state_dict().items():
        g["p." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "g8_gru.npz"), **g)
    print("codec fixtures written")


if __name__ == "__main__":
    _stub_modules()
    {"mt": gen_mt, "codec": gen_codec}[sys.argv[1]]()
