"""Mirror of the array-side helpers of mg/model/MusicTransformer/utils.py used by the hot path."""
from __future__ import annotations

import os

import torch


def find_files_by_extensions(root, exts=()):
    """Every file below ``root`` whose lower-cased name ends with one of ``exts`` (all files when ``exts`` is empty), in
    os.walk order -- the order the reference's 80/10/10 split depends on (utils.py:10-22, data.py:11-17)."""
    wanted = tuple(exts)
    for folder, _dirs, names in os.walk(root):
        for name in names:
            if not wanted or name.lower().endswith(wanted):
                yield os.path.join(folder, name)


def dict2params(d, f=','):
    return f.join('{}={}'.format(k, v) for k, v in d.items())


def params2dict(p, f=',', e='='):
    """inverse of dict2params (utils.py:38-47); values go through ast.literal_eval, not the reference's eval()"""
    import ast
    out = {}
    for field in p.split(f):
        key, sep, value = field.partition(e)
        if sep:
            out[key] = ast.literal_eval(value)
    return out


def sequence_mask(length, max_length=None):
    """[len(length), max_length] bool, row r true on columns < length[r]  (TensorFlow's sequence_mask; utils.py:183-188)"""
    width = length.max() if max_length is None else max_length
    cols = torch.arange(width, dtype=length.dtype, device=length.device)
    return cols[None, :] < length[:, None]


def get_masked_with_pad_tensor(size, src, trg, pad_token):
    """-> (src_mask, trg_mask, look_ahead_mask) as the reference returns them (utils.py:58-83): the first two are the
    Python bools of its two whole-tensor ``torch.equal`` calls (both on ``src``), the third is the [B,1,size,size] mask
    ``trg[b,j] == pad  or  j > i``.  The kernels never materialise that mask (key-padding bitmap + causal structure);
    this helper exists for API parity, for the layer-level ``forward(x, mask)`` and for tests."""
    all_pad = bool((src == pad_token).all().item()) if src.numel() else True
    if trg is None:
        return all_pad, None, None
    key_is_pad = (trg == pad_token)[:, None, None, :]                      # [B,1,1,L]
    i = torch.arange(size, device=trg.device)
    future = i[None, :] > i[:, None]                                        # [size,size]: key j after query i
    return all_pad, all_pad, key_is_pad | future[None, None]


def check_no_leading_pads(batch, pad_token) -> None:
    """Host-side guard at the data boundary (numpy / CPU tensor [B, L], before the H2D copy): a sequence must not START with
    padding.  A row whose first tokens are padding has queries with every visible key masked; there the reference's softmax is
    a rounding artefact of ``-1e9 + x`` in fp32 (layers.py:99-102) and the kernels return the uniform average over j <= i
    instead -- outside the parity contract, so such input is refused rather than trained on.  Trailing and
    interior pads are fine: every real query still sees a real key, and the kernels mask the padded keys exactly as the
    reference's look-ahead mask does (utils.py:58-83).  A row of nothing but padding has no real query and passes."""
    import numpy as np
    a = batch.numpy() if isinstance(batch, torch.Tensor) else np.asarray(batch)
    if a.ndim != 2 or a.shape[1] == 0:
        return
    is_pad = (a == pad_token)
    bad = is_pad[:, 0] & ~is_pad.all(axis=1)
    if bool(bad.any()):
        rows = np.nonzero(bad)[0][:4].tolist()
        raise ValueError(f"batch rows {rows} start with padding token {pad_token} and hold real tokens later: leading padding is "
                         "outside the parity contract with the reference (fully masked queries); pads may trail or sit inside")


check_pads_trail = check_no_leading_pads      # the name of rounds 4-5, when interior pads were refused as well


def event_indeces_to_midi_file(event_indeces, midi_file_name, velocity_scale=0.8):
    """utils.py:25-31"""
    from .sequence import EventSeq
    event_seq = EventSeq.from_array(event_indeces)
    note_seq = event_seq.to_note_seq()
    for note in note_seq.notes:
        note.velocity = int((note.velocity - 64) * velocity_scale + 64)
    note_seq.to_midi_file(midi_file_name)
    return len(note_seq.notes)
