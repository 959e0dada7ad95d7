"""Token-id layout shared by the three event codecs (sequence.EventSeq, REMI_EventSeq, MuMIDI_EventSeq).

A codec's vocabulary is an ordered list of (feature name, number of slots); ids are handed out feature after feature
with no gaps.  The reference spells the same arithmetic out once per codec (utils/sequence.py:204-221, utils/REMI.py:434-474,
utils/MuMIDI.py:352-405); here it lives in one place and the codecs only state their layouts."""
from __future__ import annotations

import collections
from itertools import accumulate
from typing import Callable, Iterable, Optional, Sequence, Tuple

import numpy as np


def slots(layout: Iterable[Tuple[str, int]]) -> "collections.OrderedDict[str, int]":
    """feature -> number of ids, in vocabulary order"""
    return collections.OrderedDict(layout)


def id_ranges(layout: Iterable[Tuple[str, int]]) -> "collections.OrderedDict[str, range]":
    """feature -> range of its ids (consecutive features are adjacent)"""
    names, sizes = zip(*layout)
    ends = list(accumulate(sizes))
    return collections.OrderedDict((n, range(e - s, e)) for n, s, e in zip(names, sizes, ends))


def id_table(ranges: "collections.OrderedDict[str, range]",
             label: Optional[Callable[[str, int], str]] = None) -> "collections.OrderedDict[int, Tuple[str, int]]":
    """id -> (feature name, value inside the feature); ``label(feature, value)`` may substitute the reported name"""
    table = collections.OrderedDict()
    for name, ids in ranges.items():
        table.update((i, (label(name, v) if label else name, v)) for v, i in enumerate(ids))
    return table


def word_dtype(vocab_size: int):
    """ids are stored as uint8 while they fit, uint16 beyond (the reference's on-disk choice)"""
    return np.uint8 if vocab_size <= 256 else np.uint16


def encode(pairs: Sequence[Tuple[range, int]], vocab_size: int) -> np.ndarray:
    """[(feature id range, value)] -> id array.  ``range.__getitem__`` raises IndexError for a value the feature has no
    slot for -- the reference's behaviour (e.g. REMI note_velocity >= 4), kept on purpose."""
    return np.array([ids[v] for ids, v in pairs], dtype=word_dtype(vocab_size))
