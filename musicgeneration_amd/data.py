"""Mirror of mg/model/MusicTransformer/data.py: ``Data(dir_path, max_length)`` with
``batch / slide_seq2seq_batch / seq2seq_batch / smallest_encoder_batch`` and ``file_dict``.

File format: each ``*.data`` file is ``torch.save(np.ndarray[uint8|uint16, T])`` written
by the reference's preprocess_*.py.  The reference unpickles every sampled file on every step
(data.py:96-107); here every file is loaded ONCE into host memory at construction (so the GPU is
not starved), and each rank may use its own ``random.Random`` stream (data-parallel sharding).
MuMIDI files hold a dict {'melody','arrangement'}; the reference's filter drops them
(len(dict) == 2 < max_length) -- pass ``field='melody'`` to train on one of the two arrays."""
from __future__ import annotations

import random
from typing import Dict, List, Optional

import numpy as np
import torch

from . import utils


def _load_array(fname, field=None):
    obj = torch.load(fname, weights_only=False)
    if isinstance(obj, dict):
        if field is None:
            return None, len(obj)
        obj = obj[field]
    arr = np.asarray(obj)
    return arr, len(arr)


class Data:
    def __init__(self, dir_path, max_length, field: Optional[str] = None, rng: Optional[random.Random] = None,
                 min_length: Optional[int] = None):
        """``min_length`` (extra): keep only files with at least this many events.  The reference keeps
        ``len >= max_length`` (data.py:33-40) although ``slide_seq2seq_batch`` crops ``max_length + 1`` events with
        ``randrange(0, len - (max_length+1))``: a file of exactly ``max_length`` events raises IndexError at sampling
        time and one of ``max_length + 1`` raises ValueError (empty range).  Data-parallel training passes
        ``min_length = max_length + 2`` so that no rank can ever skip a micro-batch on its own (every rank must
        issue the same collectives)."""
        self.files = list(utils.find_files_by_extensions(dir_path, ['.data']))
        self.field = field
        self.min_length = min_length
        self._rng = rng if rng is not None else random
        self._cache: Dict[str, np.ndarray] = {}
        n = len(self.files)
        self.file_dict = {
            'train': self.file_filter(self.files[:int(n * 0.8)], max_length),
            'valid': self.file_filter(self.files[int(n * 0.8): int(n * 0.9)], max_length),
            'test': self.file_filter(self.files[int(n * 0.9):], max_length),
        }
        self._seq_file_name_idx = 0
        self._seq_idx = 0

    def __repr__(self):
        return (f"<class Data has train: {len(self.file_dict['train'])}, val: {len(self.file_dict['valid'])},"
                f"test: {len(self.file_dict['test'])} files>")

    def file_filter(self, files, max_length):
        """keep files with len(data) >= max_length (data.py:33-40); arrays stay cached in RAM"""
        kept = []
        for fname in files:
            arr, n = _load_array(fname, self.field)
            if arr is not None and max(max_length, self.min_length or 0) <= n:
                self._cache[fname] = arr
                kept.append(fname)
        return kept

    def check_vocab(self, vocab_size):
        """Raise if any kept file holds a token id outside [0, vocab_size): the reference's nn.Embedding / one_hot raise
        on such input (a dataset / --repr mismatch), the HIP kernels would clamp silently.  Host-side, once per dataset."""
        for fname, arr in self._cache.items():
            if len(arr) and (int(arr.max()) >= vocab_size or int(arr.min()) < 0):
                raise ValueError(f"{fname}: token id {int(arr.max())} outside the vocabulary [0, {vocab_size}) "
                                 f"(dataset written for another event representation?)")

    def _get_seq(self, fname, max_length=None):
        """random crop of max_length events (data.py:96-107); IndexError when the file is too short"""
        data = self._cache.get(fname)
        if data is None:
            data, _ = _load_array(fname, self.field)
            self._cache[fname] = data
        if max_length is not None:
            if max_length <= len(data):
                start = self._rng.randrange(0, len(data) - max_length)
                data = data[start:start + max_length]
            else:
                raise IndexError
        return data

    def batch(self, batch_size, length, mode='train'):
        batch_files = self._rng.sample(self.file_dict[mode], k=batch_size)
        batch_data = [self._get_seq(file, length) for file in batch_files]
        return np.array(batch_data, dtype=np.int16)

    def seq2seq_batch(self, batch_size, length, mode='train'):
        data = self.batch(batch_size, length * 2, mode)
        return data[:, :length], data[:, length:]

    def smallest_encoder_batch(self, batch_size, length, mode='train'):
        data = self.batch(batch_size, length * 2, mode)
        return data[:, :length // 100], data[:, length // 100:length // 100 + length]

    def slide_seq2seq_batch(self, batch_size, length, mode='train'):
        data = self.batch(batch_size, length + 1, mode)
        return data[:, :-1], data[:, 1:]


# --------------------------------------------------------------------------------------------------
# GRU feeder: mirror of mg/model/utils/data.py:23-47 (SeqBatchify, MyDataset) and :49-123 (Event_Dataset)
# --------------------------------------------------------------------------------------------------
def flatten_padded_sequences(outs, lengths):
    """[B, mx, V] batch-first outputs -> the rows that have a label, concatenated: outs[i, :lengths[i]-1] for every i
    (utils/data.py:14-21); pairs with SeqBatchify's Y."""
    if lengths is None:
        return outs.contiguous().view(-1, outs.shape[-1])
    return torch.cat([outs[i, :int(lengths[i]) - 1] for i in range(outs.shape[0])], 0)


def SeqBatchify(inputs):
    """sort by length (desc), zero-pad to the longest -> (X int16 [B,Tmax], Y = concat of X[i,1:len_i], lengths)"""
    inputs = sorted(inputs, key=lambda i: len(i), reverse=True)
    lengths = np.array([len(item) for item in inputs])
    mx_length = np.max(lengths)
    X = np.zeros((len(inputs), mx_length), dtype=np.int16)
    for i in range(len(inputs)):
        X[i, :lengths[i]] = np.array(inputs[i])
    Y = np.concatenate([np.array(X[i])[1:lengths[i]] for i in range(len(inputs))])
    return X, Y, lengths


class MyDataset(torch.utils.data.Dataset):
    def __init__(self, seqs):
        self.seqs = seqs

    def __getitem__(self, index):
        return self.seqs[index]

    def __len__(self):
        return len(self.seqs)


class Event_Dataset:
    """All ``*.data`` arrays with len >= limlen, held in RAM; window index + time-major collate."""

    def __init__(self, root, limlen=None, verbose=False):
        import os
        assert os.path.isdir(root), root
        self.root = root
        self.samples = []
        self.seqlens = []
        for path in utils.find_files_by_extensions(root, ['.data']):
            eventseq, n = _load_array(path)
            if eventseq is not None and n >= (limlen or 0):
                self.samples.append(eventseq)
                self.seqlens.append(n)
        self.avglen = np.mean(self.seqlens) if self.seqlens else 0.0

    def count(self, v):
        a = sorted(self.seqlens)
        x = int(np.searchsorted(a, v, side='left'))
        return 100 * x / len(a)

    def batches(self, batch_size, window_size, stride_size):
        """list of (file index, (start, end)) windows            utils/data.py:74-78"""
        return [(i, (j, j + window_size))
                for i, seqlen in enumerate(self.seqlens)
                for j in range(0, seqlen - window_size, stride_size)]

    def SegBatchify(self, data):
        """collate -> np [T, B] (time-major)                      utils/data.py:104-114"""
        return np.stack([self.samples[i][start:end] for i, (start, end) in data], axis=1)

    Batchify = SegBatchify

    def __repr__(self):
        return (f'Dataset(root="{self.root}", samples={len(self.samples)}, avglen={self.avglen})')
