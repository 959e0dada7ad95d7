"""Host-side logic that needs no GPU: schedule, data feeder, CLI surface, sampling filter, codecs' feeder."""
import json
import os

import numpy as np
import pytest
import torch


def test_custom_schedule_matches_reference(golden_dir):
    from musicgeneration_amd.criterion import CustomSchedule
    with open(os.path.join(golden_dir, "g4_schedule.json")) as f:
        g = json.load(f)
    for dm, vals in g["g4"].items():
        sch = CustomSchedule(int(dm))
        for s, v in zip(g["g4_steps"], vals):
            assert abs(sch.rate(s) - v) <= 1e-12 * abs(v)

    class Opt:
        param_groups = [{"lr": 0.0}]
        n = 0

        def step(self):
            self.n += 1
    o = Opt()
    sch = CustomSchedule(256, optimizer=o)
    sch.step(); sch.step()
    assert o.n == 2 and o.param_groups[0]["lr"] == sch.rate(2) == sch._rate


def _make_dataset(tmp, n=10, lens=(40, 80, 20)):
    rng = np.random.RandomState(0)
    for i in range(n):
        arr = rng.randint(0, 308, size=lens[i % len(lens)]).astype(np.uint16)
        torch.save(arr, os.path.join(tmp, f"piece{i:02d}-abcdef.data"))
    torch.save({"melody": np.arange(50, dtype=np.uint16), "arrangement": np.arange(60, dtype=np.uint16)},
               os.path.join(tmp, "mumidi-0.data"))


def test_data_feeder(tmp_path):
    import random
    from musicgeneration_amd.data import Data
    _make_dataset(str(tmp_path))
    ds = Data(str(tmp_path), 33, rng=random.Random(0))
    total = sum(len(v) for v in ds.file_dict.values())
    assert total == 7            # the 20-event files and the dict file are filtered out (len < max_length)
    x, y = ds.slide_seq2seq_batch(2, 32)
    assert x.shape == y.shape == (2, 32) and x.dtype == np.int16
    assert (x[:, 1:] == y[:, :-1]).all()          # y is x shifted by one event
    with pytest.raises(IndexError):
        ds._get_seq(ds.file_dict["train"][0], 10 ** 6)
    with pytest.raises(ValueError):
        ds.batch(100, 16)                          # random.sample: more files requested than available
    d2 = Data(str(tmp_path), 40, field="melody")
    assert any("mumidi" in f for fs in d2.file_dict.values() for f in fs)
    # two ranks with different streams draw different crops; same seed reproduces
    a = Data(str(tmp_path), 33, rng=random.Random(1)).slide_seq2seq_batch(2, 32)[0]
    b = Data(str(tmp_path), 33, rng=random.Random(1)).slide_seq2seq_batch(2, 32)[0]
    c = Data(str(tmp_path), 33, rng=random.Random(2)).slide_seq2seq_batch(2, 32)[0]
    assert (a == b).all() and not (a == c).all()


def test_train_cli_surface_matches_reference():
    from musicgeneration_amd import config
    from musicgeneration_amd.train import get_options, vocab_of
    o = get_options([])
    assert (o.epochs, o.batch_size, o.max_seq, o.multi_gpu, o.load_path) == (config.epochs, 6, 2048, 'False', None)
    o = get_options("-s out/ -d data/ -e 3 -i 7 -b 4 -l 0.01 -w 128 -S 5 -g True -m ck.pth -M 512".split())
    assert (o.save_path, o.data_path, o.epochs, o.saving_interval, o.batch_size, o.learning_rate, o.window_size,
            o.stride_size, o.multi_gpu, o.load_path, o.max_seq) == ("out/", "data/", 3, 7, 4, 0.01, 128, 5, "True",
                                                                    "ck.pth", 512)
    o = get_options(["--save_path", "a", "--dataset", "b", "--epochs", "1", "--saving-interval", "2", "--batch-size", "3",
                     "--learning-rate", "0.5", "--window-size", "4", "--stride-size", "5", "--multi_gpu", "False",
                     "--load_path", "c", "--max_seq", "64"])
    assert o.max_seq == 64 and o.save_path == "a"
    assert (vocab_of("midi_like"), vocab_of("remi"), vocab_of("mumidi")) == (309, 337, 486)
    assert config.pad_token == 308 and config.vocab_size == 309 and config.accum_grad == 12 and config.label_smooth == 0.1


def test_filter_probs():
    from musicgeneration_amd.network import filter_probs
    p = torch.tensor([[0.5, 0.3, 0.15, 0.05], [0.25, 0.25, 0.25, 0.25]])
    assert torch.allclose(filter_probs(p), p)                       # defaults: the reference's full softmax
    k2 = filter_probs(p, top_k=2)
    assert torch.allclose(k2[0], torch.tensor([0.625, 0.375, 0, 0]))
    n = filter_probs(p, top_p=0.7)          # smallest prefix whose mass reaches 0.7: {0.5, 0.3}
    assert torch.allclose(n[0], torch.tensor([0.5, 0.3, 0, 0]) / 0.8)
    assert torch.allclose(filter_probs(p, top_p=0.9)[0], torch.tensor([0.5, 0.3, 0.15, 0]) / 0.95)
    t = filter_probs(p, temperature=0.5)
    assert t[0, 0] > p[0, 0] and abs(t.sum(-1) - 1).max() < 1e-6


def test_params2dict_no_eval():
    from musicgeneration_amd.utils import dict2params, params2dict
    assert params2dict("a=1,b=0.5,c='x'") == {"a": 1, "b": 0.5, "c": "x"}
    assert dict2params({"a": 1, "b": 2}) == "a=1,b=2"
    with pytest.raises(Exception):
        params2dict("a=__import__('os').getcwd()")      # the reference would eval this


def test_mask_helper_matches_golden(golden_dir):
    from musicgeneration_amd.utils import get_masked_with_pad_tensor
    g = dict(np.load(os.path.join(golden_dir, "g6_mask_pe.npz")))
    x = torch.from_numpy(g["x"])
    _, _, m = get_masked_with_pad_tensor(8, x, x, int(g["pad"]))
    assert (m.numpy() == g["mask"]).all()
    from musicgeneration_amd.layers import sinusoid
    np.testing.assert_allclose(sinusoid(8, 16), g["pe"], rtol=0, atol=1e-12)


def test_event_dataset_and_seqbatchify(tmp_path):
    from musicgeneration_amd.data import Event_Dataset, MyDataset, SeqBatchify
    _make_dataset(str(tmp_path))
    ds = Event_Dataset(str(tmp_path), limlen=40)
    assert len(ds.samples) == 7 and min(ds.seqlens) == 40           # the dict file and the short files are skipped
    idx = ds.batches(4, 16, 8)
    assert all(e - s == 16 for _, (s, e) in idx) and len(idx) == 4 * 3 + 3 * 8
    batch = ds.SegBatchify(idx[:5])
    assert batch.shape == (16, 5)
    assert (batch[:, 0] == ds.samples[idx[0][0]][idx[0][1][0]:idx[0][1][1]]).all()
    assert len(MyDataset(idx)) == len(idx) and MyDataset(idx)[3] == idx[3]
    X, Y, lengths = SeqBatchify([[1, 2, 3], [4, 5, 6, 7, 8], [9]])
    assert lengths.tolist() == [5, 3, 1] and X.shape == (3, 5) and X.dtype == np.int16
    assert Y.tolist() == [5, 6, 7, 8, 2, 3]


# ---- G10: the integer feeders against outputs of the reference itself (tests/golden/gen_golden.py mt2 / codec) ----
def _write_feeder_corpus(root):
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(os.path.dirname(__file__), "golden", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)            # only defines functions; the reference is NOT imported by this
    gg.write_corpus(root)
    return gg.feeder_corpus()


def test_g10a_data_feeder_bit_exact(golden_dir, tmp_path, monkeypatch):
    """Data.file_filter / split / _get_seq / slide_seq2seq_batch / seq2seq_batch reproduce the reference's
    output for the same `random` seed, bit for bit (data.py:11-17,33-49,51-67,96-107)."""
    import random
    from musicgeneration_amd import data as D
    g = dict(np.load(os.path.join(golden_dir, "g10a_data_feeder.npz")))
    corpus = _write_feeder_corpus(str(tmp_path))
    names = [str(n) for n in g["names"]]
    # the 80/10/10 split follows directory-walk order, a property of the file system: pinned as in the generator
    monkeypatch.setattr(D.utils, "find_files_by_extensions", lambda r, exts=None: iter([os.path.join(r, n) for n in names]))
    L = int(g["max_length"])
    dd = D.Data(str(tmp_path), L)          # module-level `random`, like the reference
    for mode in ("train", "valid", "test"):
        assert [os.path.basename(f) for f in dd.file_dict[mode]] == [str(n) for n in g["files_" + mode]], mode
    outcomes = []
    for n in names:
        try:
            random.seed(1)
            dd._get_seq(os.path.join(str(tmp_path), n), L + 1)
            outcomes.append("ok")
        except Exception as e:      # noqa: BLE001
            outcomes.append(type(e).__name__)
    assert outcomes == [str(o) for o in g["crop_outcome"]]          # IndexError / ValueError / ok, file by file
    # the DP-safe filter keeps exactly the files whose crop cannot fail
    safe = D.Data(str(tmp_path), L, min_length=L + 2)
    assert [os.path.basename(f) for f in safe.file_dict["train"]] == [str(n) for n in g["files_train_safe"]]
    dd.file_dict["train"] = safe.file_dict["train"]
    for seed in (0, 7):
        random.seed(seed)
        for k in range(3):
            x, y = dd.slide_seq2seq_batch(4, L)
            assert x.dtype == g[f"x_{seed}_{k}"].dtype == np.int16
            assert (x == g[f"x_{seed}_{k}"]).all() and (y == g[f"y_{seed}_{k}"]).all(), (seed, k)
    random.seed(3)
    a, b = dd.seq2seq_batch(2, 16)
    assert (a == g["s2s_x"]).all() and (b == g["s2s_y"]).all()
    # a private random.Random stream seeded alike draws the same batches (what each DP rank uses)
    dr = D.Data(str(tmp_path), L, rng=random.Random(7), min_length=L + 2)
    assert (dr.slide_seq2seq_batch(4, L)[0] == g["x_7_0"]).all()
    assert len(corpus) == len(names)


def test_g10b_event_dataset_bit_exact(golden_dir, tmp_path, monkeypatch):
    """Event_Dataset.batches / SegBatchify / count and SeqBatchify vs the reference (utils/data.py:23-36,49-114)."""
    from musicgeneration_amd import data as D
    g = dict(np.load(os.path.join(golden_dir, "g10b_gru_feeders.npz")))
    _write_feeder_corpus(str(tmp_path))
    names = [str(n) for n in g["names"]]
    monkeypatch.setattr(D.utils, "find_files_by_extensions", lambda r, exts=None: iter([os.path.join(r, n) for n in names]))
    ds = D.Event_Dataset(str(tmp_path), limlen=int(g["limlen"]))
    assert ds.seqlens == g["seqlens"].tolist() and ds.avglen == float(g["avglen"])
    idx = ds.batches(4, 16, 8)
    assert np.array([[i, s, e] for i, (s, e) in idx], dtype=np.int64).tolist() == g["batches"].tolist()
    pick = [(int(i), (int(s), int(e))) for i, s, e in g["pick"]]
    seg = ds.SegBatchify(pick)
    assert seg.dtype == g["seg"].dtype and (seg == g["seg"]).all()
    assert (ds.Batchify(pick) == g["seg"]).all()
    assert ds.count(50) == float(g["count50"])
    ragged, o = [], 0
    for n in g["sb_in_lens"]:
        ragged.append(g["sb_in"][o:o + int(n)])
        o += int(n)
    X, Y, lengths = D.SeqBatchify(ragged)
    assert X.dtype == g["sb_X"].dtype and (X == g["sb_X"]).all()
    assert Y.dtype == g["sb_Y"].dtype and (Y == g["sb_Y"]).all() and (lengths == g["sb_lengths"]).all()


def test_data_check_vocab_rejects_out_of_range_ids(tmp_path):
    from musicgeneration_amd.data import Data
    torch.save(np.array([1, 2, 400] * 20, dtype=np.uint16), os.path.join(str(tmp_path), "a-0.data"))
    torch.save(np.array([1, 2, 3] * 20, dtype=np.uint16), os.path.join(str(tmp_path), "b-1.data"))
    ds = Data(str(tmp_path), 16)
    ds.check_vocab(401)
    with pytest.raises(ValueError, match="outside the vocabulary"):
        ds.check_vocab(309)


def test_leading_pads_are_refused_and_trailing_or_interior_pads_accepted():
    """The data-boundary guard behind DESIGN.md section 5: only rows that START with padding (and hold real tokens later) have
    fully masked queries -- outside the parity contract (the reference's result there is a rounding artefact, fixture
    g2b_leadpad) -- and are refused before they reach the GPU.  Trailing and interior pads are what the reference's look-ahead
    mask handles (utils.py:58-83) and pass, as does a row of nothing but padding (no real query)."""
    import numpy as np
    import pytest
    from musicgeneration_amd import utils
    pad = 9
    ok = np.array([[1, 2, 3, 9, 9], [4, 5, 6, 7, 8], [9, 9, 9, 9, 9], [1, 9, 2, 9, 9], [1, 2, 9, 3, 4]], dtype=np.int16)
    utils.check_no_leading_pads(ok, pad)
    utils.check_no_leading_pads(torch.from_numpy(ok.astype(np.int64)), pad)
    utils.check_pads_trail(ok, pad)                        # the old name is an alias
    for bad in ([[9, 1, 2, 3, 4]], [[9, 9, 2, 9, 9]], [[1, 2, 3, 4, 5], [9, 9, 9, 9, 1]]):
        with pytest.raises(ValueError, match="leading padding"):
            utils.check_no_leading_pads(np.array(bad, dtype=np.int16), pad)


def test_pack_frag_layout_matches_the_header():
    """ops.pack_frag (pure torch, runs without a GPU): unit ((nt*K/16 + ks)*64 + lane) of the packed buffer holds
    W[32 nt + lane % 32][16 ks + 8 (lane // 32) .. +7] -- the layout include/mgx.h documents for the *_frag / GRU entry points."""
    import torch
    from musicgeneration_amd import ops
    N, K = 96, 64
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251          # exactly representable in bf16
    packed = ops.pack_frag(w).reshape(-1, 8).float()
    for nt in range(N // 32):
        for ks in range(K // 16):
            for lane in (0, 5, 31, 32, 47, 63):
                unit = (nt * (K // 16) + ks) * 64 + lane
                want = w[32 * nt + lane % 32, 16 * ks + 8 * (lane // 32): 16 * ks + 8 * (lane // 32) + 8]
                assert torch.equal(packed[unit], want)
    import pytest
    with pytest.raises(ValueError):
        ops.pack_frag(torch.zeros(40, 64))


def test_dropout_seed_depends_on_the_data_parallel_rank():
    """every rank of a DP job usually calls torch.manual_seed with the same value; the dropout seed must still differ per rank
    (identical masks on different rows are a correlated regulariser), and stay a pure function of (seed, rank, call number)"""
    from musicgeneration_amd.network import MusicTransformer
    torch.manual_seed(0)
    seeds = {}
    for rank in (None, 0, 1, 2):
        mt = MusicTransformer(embedding_dim=64, vocab_size=20, num_layer=1, max_seq=32)
        if rank is not None:
            mt._dp = type("DP", (), {"rank": rank})()
        seeds[rank] = [mt._next_seed() for _ in range(3)]
    assert seeds[None] == seeds[0]                       # no DP == rank 0
    flat = seeds[0] + seeds[1] + seeds[2]
    assert len(set(flat)) == len(flat)
