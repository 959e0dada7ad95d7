"""REMI event codec (array side), mirroring mg/model/utils/REMI.py:404-536.

Vocabulary (336 ids): note_on 0-126, note_duration 127-190, note_velocity 191-194, bar 195,
position 196-211, tempo_class 212-214, tempo_value 215-274, chord 275-335 (chord values are the
strings 'C:maj' ... 'N:N').  Reference quirk kept: note_velocity has only 4 slots, so
``to_array`` raises IndexError for a velocity value >= 4 (REMI.py:452 vs :206-209)."""
from __future__ import annotations


import numpy as np

from . import _vocab

DEFAULT_FRACTION = 16
DEFAULT_DURATION_RANGE = range(60, 3841)
DEFAULT_DURATION_STEP = 60
DEFAULT_DURATION_BINS = np.arange(DEFAULT_DURATION_RANGE.start, DEFAULT_DURATION_RANGE.stop,
                                  DEFAULT_DURATION_STEP, dtype=int)
DEFAULT_tempo_INTERVALS = [range(30, 90), range(90, 150), range(150, 210)]
DEFAULT_VELOCITY = 100
DEFAULT_PITCH_RANGE = range(0, 127)
DEFAULT_VELOCITY_STEPS = 4
DEFAULT_VELOCITY_RANGE = range(DEFAULT_VELOCITY_STEPS, 128)
DEFAULT_VELOCITY_BINS = np.arange(DEFAULT_VELOCITY_RANGE.start, DEFAULT_VELOCITY_RANGE.stop, DEFAULT_VELOCITY_STEPS)
DEFAULT_RESOLUTION = 480

chord_quality = ['maj', 'min', 'dim', 'aug', 'dom']
chord_root = ['C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#', 'A', 'A#', 'B']
chord_map = {}
for _qi, _q in enumerate(chord_quality):
    for _ri, _r in enumerate(chord_root):
        chord_map[_r + ':' + _q] = _qi * len(chord_root) + _ri
chord_map['N:N'] = len(chord_quality) * len(chord_root)
inv_chord_map = {v: k for k, v in chord_map.items()}


class Event(object):
    def __init__(self, name, time, value, text):
        self.name = name
        self.time = time
        self.value = value
        self.text = text

    def __repr__(self):
        return 'Event(name={}, time={}, value={}, text={})'.format(self.name, self.time, self.value, self.text)


class REMI_EventSeq:
    pitch_range = DEFAULT_PITCH_RANGE
    velocity_range = DEFAULT_VELOCITY_RANGE
    velocity_steps = DEFAULT_VELOCITY_STEPS
    duration_bins = DEFAULT_DURATION_BINS

    def __init__(self, events=[]):
        pass

    @staticmethod
    def _layout():
        """(feature, slots) in id order -- REMI.py:434-460.  Read from the class attributes on every call, so a caller that
        narrows ``pitch_range`` / ``duration_bins`` gets the vocabulary the reference would give."""
        c = REMI_EventSeq
        return (('note_on', len(c.pitch_range)), ('note_duration', len(c.duration_bins)),
                ('note_velocity', c.velocity_steps), ('bar', 1), ('position', DEFAULT_FRACTION),
                ('tempo_class', len(DEFAULT_tempo_INTERVALS)), ('tempo_value', len(DEFAULT_tempo_INTERVALS[0])),
                ('chord', len(chord_map)))

    @staticmethod
    def feat_dims():
        return _vocab.slots(REMI_EventSeq._layout())

    @staticmethod
    def dim():
        return sum(n for _, n in REMI_EventSeq._layout())

    @staticmethod
    def feat_ranges():
        return _vocab.id_ranges(REMI_EventSeq._layout())

    @staticmethod
    def dims_feat():
        """id -> (feature name, value)          REMI.py:462-474"""
        return _vocab.id_table(REMI_EventSeq.feat_ranges())

    @staticmethod
    def write_midi(events, output_path, prompt_path=None):
        """utils/REMI.py:538-672 without the prompt branch: scan the event list for the patterns
        (position, note_velocity, note_on, note_duration), (position, chord), (position, tempo_class, tempo_value) and
        'bar'; place them on a 4/4 grid of DEFAULT_FRACTION positions per bar at DEFAULT_RESOLUTION ticks per beat; write
        notes, tempo changes and chord markers.  The reference writes through miditoolkit; here the built-in SMF writer
        produces the same content (smf.write_ticks).  Returns (notes, tempos, chords) in ticks."""
        if prompt_path is not None:
            raise NotImplementedError("continuing a prompt MIDI file needs miditoolkit's parser")
        from . import smf
        temp_notes, temp_chords, temp_tempos = [], [], []
        ev = events
        for i in range(len(ev) - 3):
            if ev[i].name == 'bar' and i > 0:
                temp_notes.append('bar'); temp_chords.append('bar'); temp_tempos.append('bar')
            elif ev[i].name == 'position' and ev[i + 1].name == 'note_velocity' and ev[i + 2].name == 'note_on' and \
                    ev[i + 3].name == 'note_duration':
                temp_notes.append([int(ev[i].value), int(DEFAULT_VELOCITY_BINS[int(ev[i + 1].value)]), int(ev[i + 2].value),
                                   int(DEFAULT_DURATION_BINS[int(ev[i + 3].value)])])
            elif ev[i].name == 'position' and ev[i + 1].name == 'chord':
                temp_chords.append([int(ev[i].value), ev[i + 1].value])
            elif ev[i].name == 'position' and ev[i + 1].name == 'tempo_class' and ev[i + 2].name == 'tempo_value':
                temp_tempos.append([int(ev[i].value), DEFAULT_tempo_INTERVALS[ev[i + 1].value].start + int(ev[i + 2].value)])
        ticks_per_bar = DEFAULT_RESOLUTION * 4

        def on_grid(items):
            out, bar = [], 0
            for it in items:
                if it == 'bar':
                    bar += 1
                    continue
                flags = np.linspace(bar * ticks_per_bar, (bar + 1) * ticks_per_bar, DEFAULT_FRACTION, endpoint=False, dtype=int)
                out.append([int(flags[it[0]])] + list(it[1:]))
            return out

        notes = [(vel, pitch, st, st + dur) for st, vel, pitch, dur in on_grid(temp_notes)]
        chords = on_grid(temp_chords)
        tempos = on_grid(temp_tempos)
        smf.write_ticks(output_path, notes, DEFAULT_RESOLUTION, [(st, bpm) for st, bpm in tempos],
                        [(st, text) for st, text in chords], program=0)
        return notes, tempos, chords

    @staticmethod
    def next_token_table(pad: bool = True):
        """First-order grammar of a REMI stream as ``write_midi`` reads it (utils/REMI.py:549-581): bar -> position;
        position -> note_velocity | chord | tempo_class; note_velocity -> note_on -> note_duration; tempo_class ->
        tempo_value; note_duration | chord | tempo_value -> position | bar.  Returns np.uint32 [V, ceil(V/32)], bit v of row
        t set iff token v may follow token t (V = dim() + 1 with the pad id, which may follow nothing and is followed by
        anything but itself).  Feed it to ``MusicTransformer.generate_cached(grammar=...)``: the mask is applied inside the
        sampling kernel, raising the share of generated tokens that decode to notes."""
        fr = REMI_EventSeq.feat_ranges()
        V = REMI_EventSeq.dim() + (1 if pad else 0)
        W = (V + 31) // 32
        follow = {'bar': ['position'], 'position': ['note_velocity', 'chord', 'tempo_class'], 'note_velocity': ['note_on'],
                  'note_on': ['note_duration'], 'note_duration': ['position', 'bar'], 'chord': ['position', 'bar'],
                  'tempo_class': ['tempo_value'], 'tempo_value': ['position', 'bar']}
        table = np.zeros((V, W), dtype=np.uint32)

        def allow(row, names):
            for nm in names:
                for v in fr[nm]:
                    table[row, v >> 5] |= np.uint32(1 << (v & 31))

        for nm, rng in fr.items():
            for t in rng:
                allow(t, follow[nm])
        if pad:
            allow(V - 1, list(fr.keys()))
        return table

    @staticmethod
    def get_velocity_bins():
        n = REMI_EventSeq.velocity_range.stop - REMI_EventSeq.velocity_range.start
        return np.arange(REMI_EventSeq.velocity_range.start, REMI_EventSeq.velocity_range.stop,
                         n / (REMI_EventSeq.velocity_steps - 1))

    @staticmethod
    def to_array(events):
        """events -> ids (REMI.py:510-520); a chord's value is its name, every other value indexes the feature's slots"""
        ids = REMI_EventSeq.feat_ranges()
        return _vocab.encode([(ids[e.name], chord_map[e.value] if e.name == 'chord' else e.value) for e in events],
                             REMI_EventSeq.dim())

    @staticmethod
    def to_event(words):
        """ids -> events without times (REMI.py:522-531)"""
        table = REMI_EventSeq.dims_feat()
        named = (table[int(w)] for w in words)
        return [Event(name, None, inv_chord_map[v] if name == 'chord' else v, None) for name, v in named]

    @staticmethod
    def from_array(words):
        return REMI_EventSeq.to_event(words)
