"""Mirror of the array-side helpers of mg/model/MusicTransformer/utils.py used by the hot path."""
from __future__ import annotations

import os

import torch


def find_files_by_extensions(root, exts=[]):
    """utils.py:10-22"""
    def _has_ext(name):
        if not exts:
            return True
        name = name.lower()
        return any(name.endswith(ext) for ext in exts)
    for path, _, files in os.walk(root):
        for name in files:
            if _has_ext(name):
                yield os.path.join(path, name)


def dict2params(d, f=','):
    return f.join(f'{k}={v}' for k, v in d.items())


def params2dict(p, f=',', e='='):
    """utils.py:38-47, with ast.literal_eval instead of the reference's eval()."""
    import ast
    d = {}
    for item in p.split(f):
        item = item.split(e)
        if len(item) < 2:
            continue
        k, *v = item
        d[k] = ast.literal_eval('='.join(v))
    return d


def sequence_mask(length, max_length=None):
    """utils.py:183-188"""
    if max_length is None:
        max_length = length.max()
    x = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


def get_masked_with_pad_tensor(size, src, trg, pad_token):
    """utils.py:58-83: (src_mask, trg_mask, look_ahead_mask).  The kernels never materialise the
    [B,1,L,L] mask (they use a key-padding bitmap + the causal structure); this helper exists for API
    parity and for tests."""
    src = src[:, None, None, :]
    trg = trg[:, None, None, :]
    src_pad_tensor = torch.ones_like(src) * pad_token
    src_mask = torch.equal(src, src_pad_tensor)
    trg_mask = torch.equal(src, src_pad_tensor)
    dec_trg_mask = trg == torch.ones_like(trg) * pad_token
    seq_mask = ~sequence_mask(torch.arange(1, size + 1).to(trg.device), size)
    look_ahead_mask = dec_trg_mask | seq_mask
    return src_mask, trg_mask, look_ahead_mask


def event_indeces_to_midi_file(event_indeces, midi_file_name, velocity_scale=0.8):
    """utils.py:25-31"""
    from .sequence import EventSeq
    event_seq = EventSeq.from_array(event_indeces)
    note_seq = event_seq.to_note_seq()
    for note in note_seq.notes:
        note.velocity = int((note.velocity - 64) * velocity_scale + 64)
    note_seq.to_midi_file(midi_file_name)
    return len(note_seq.notes)
