"""MuMIDI multi-track event codec (array side), mirroring mg/model/utils/MuMIDI.py:340-431,543-574.

Vocabulary (485 ids): empty 0, note_on 1-256 (drums +128), note_duration 257-288,
note_velocity 289-320, bar 321, position 322-354, track 355-360, tempo_class 361-363,
tempo_value 364-423, chord 424-484.  Reference quirk kept: ``from_array`` names a track token by
its track ('melody' ... 'drum'), which ``to_array`` cannot encode (KeyError) -- MuMIDI.py:396-397."""
from __future__ import annotations


import numpy as np

from . import _vocab

DEFAULT_FRACTION = 32
DEFAULT_DURATION_STEP = 60
DEFAULT_DURATION_RANGE = range(DEFAULT_DURATION_STEP, 1921)
DEFAULT_DURATION_BINS = np.arange(DEFAULT_DURATION_RANGE.start, DEFAULT_DURATION_RANGE.stop,
                                  DEFAULT_DURATION_STEP, dtype=int)
DEFAULT_TEMPO_INTERVALS = [range(30, 90), range(90, 150), range(150, 210)]
DEFAULT_VELOCITY = 100
DEFAULT_PITCH_RANGE = range(1, 129)
DEFAULT_VELOCITY_STEPS = 4
DEFAULT_VELOCITY_RANGE = range(DEFAULT_VELOCITY_STEPS, 129)
DEFAULT_VELOCITY_BINS = np.arange(DEFAULT_VELOCITY_RANGE.start, DEFAULT_VELOCITY_RANGE.stop, DEFAULT_VELOCITY_STEPS)
DEFAULT_DRUM_TYPE = range(1, 129)
DEFAULT_RESOLUTION = 480
DEFAULT_TRACKS = ['melody', 'piano', 'bass', 'guitar', 'string', 'drum']
tracks_idx = {track: idx for idx, track in enumerate(DEFAULT_TRACKS)}
# General-MIDI programs per track (utils/MuMIDI.py:49-55); the writer uses the first of each list
instrument_numbers = {'melody': [73], 'piano': [1, 2, 3, 4, 5, 6, 7, 8], 'bass': [33, 34, 35, 36, 37, 38, 39, 40],
                      'guitar': [25, 26, 27, 28, 29, 30, 31, 32], 'drum': [114, 115, 116, 117, 118, 119], 'string': [66]}

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


class MuMIDI_EventSeq:
    pitch_range = DEFAULT_PITCH_RANGE
    velocity_range = DEFAULT_VELOCITY_RANGE
    velocity_steps = DEFAULT_VELOCITY_STEPS
    duration_bins = DEFAULT_DURATION_BINS
    feats_ranges = None
    idxs_feats = None

    def __init__(self, events=[]):
        pass

    @staticmethod
    def _layout():
        """(feature, slots) in id order -- MuMIDI.py:352-386 (drum hits share note_on, after the pitches)"""
        c = MuMIDI_EventSeq
        return (('empty', 1), ('note_on', len(c.pitch_range) + len(DEFAULT_DRUM_TYPE)), ('note_duration', len(c.duration_bins)),
                ('note_velocity', len(DEFAULT_VELOCITY_BINS)), ('bar', 1), ('position', DEFAULT_FRACTION + 1),
                ('track', len(DEFAULT_TRACKS)), ('tempo_class', len(DEFAULT_TEMPO_INTERVALS)),
                ('tempo_value', len(DEFAULT_TEMPO_INTERVALS[0])), ('chord', len(chord_map)))

    @staticmethod
    def feat_dims():
        return _vocab.slots(MuMIDI_EventSeq._layout())

    @staticmethod
    def dim():
        return sum(n for _, n in MuMIDI_EventSeq._layout())

    @staticmethod
    def feat_ranges():
        """computed once and kept in ``feats_ranges``, as the reference does (MuMIDI.py:376-386)"""
        c = MuMIDI_EventSeq
        if c.feats_ranges is None:
            c.feats_ranges = _vocab.id_ranges(c._layout())
        return c.feats_ranges

    @staticmethod
    def dims_feat():
        """id -> (name, value), kept in ``idxs_feats``; a track id reports the TRACK's name, not 'track' (MuMIDI.py:388-405)"""
        c = MuMIDI_EventSeq
        if c.idxs_feats is None:
            c.idxs_feats = _vocab.id_table(c.feat_ranges(), lambda feat, v: DEFAULT_TRACKS[v] if feat == 'track' else feat)
        return c.idxs_feats

    @staticmethod
    def get_track_id(track_name):
        return MuMIDI_EventSeq.feat_ranges()['track'][0] + tracks_idx[track_name]

    @staticmethod
    def check(feat_name, idx):
        return idx in MuMIDI_EventSeq.feat_ranges()[feat_name]

    @staticmethod
    def to_array(events):
        """events -> ids (MuMIDI.py:543-556): 'track_<name>' events are filed under 'track' (their first five characters),
        a chord's value is its name"""
        ids = MuMIDI_EventSeq.feat_ranges()

        def slot(e):
            if e.name == 'chord':
                return ids['chord'], chord_map[e.value]
            return ids[e.name[:5] if e.name.startswith('track') else e.name], e.value
        return _vocab.encode([slot(e) for e in events], MuMIDI_EventSeq.dim())

    @staticmethod
    def to_event(words):
        """ids -> events without times (MuMIDI.py:558-569).  The reference would rename a 'track' event to 'track_<name>'
        here, but its id table never reports 'track' (see dims_feat), so track events come back named after the track."""
        table = MuMIDI_EventSeq.dims_feat()
        named = (table[int(w)] for w in words)
        return [Event(name, None, inv_chord_map[v] if name == 'chord' else v, None) for name, v in named]

    @staticmethod
    def from_array(words):
        return MuMIDI_EventSeq.to_event(words)

    @staticmethod
    def next_token_table(pad: bool = True):
        """First-order grammar of a MuMIDI stream as ``write_midi`` reads it (utils/MuMIDI.py:584-622): bar -> position |
        tempo_class; tempo_class -> tempo_value -> position | bar; position -> track | chord; chord -> track | position |
        bar; track -> note_velocity -> note_on -> note_duration -> note_velocity (next note of the track) | track |
        position | bar.  'empty' and the pad id are never generated.  np.uint32 [V, ceil(V/32)] as
        ``REMI_EventSeq.next_token_table``."""
        fr = MuMIDI_EventSeq.feat_ranges()
        V = MuMIDI_EventSeq.dim() + (1 if pad else 0)
        follow = {'empty': ['bar'], 'bar': ['position', 'tempo_class'], 'tempo_class': ['tempo_value'],
                  'tempo_value': ['position', 'bar'], 'position': ['track', 'chord'], 'chord': ['track', 'position', 'bar'],
                  'track': ['note_velocity'], 'note_velocity': ['note_on'], 'note_on': ['note_duration'],
                  'note_duration': ['note_velocity', 'track', 'position', 'bar']}
        table = np.zeros((V, (V + 31) // 32), dtype=np.uint32)

        def allow(row, names):
            for nm in names:
                for v in fr[nm]:
                    table[row, v >> 5] |= np.uint32(1 << (v & 31))

        for nm, rng in fr.items():
            for t in rng:
                allow(t, follow[nm])
        if pad:
            allow(V - 1, [k for k in fr if k != 'empty'])
        return table

    @staticmethod
    def write_midi(events, output_path):
        """utils/MuMIDI.py:576-704: 'bar' advances the bar, 'position' (1-based) and a track token set the cursor,
        (note_velocity, note_on, note_duration) triples become notes of the current track (drum pitches live in the upper
        half of note_on), chords become markers, (tempo_class, tempo_value) tempo changes; 4/4 grid of DEFAULT_FRACTION
        positions per bar at DEFAULT_RESOLUTION ticks per beat; one instrument per non-empty track.
        Deviation (documented, SURVEY A14 quirk ii): the reference only recognises track tokens named ``track_<name>``,
        which its own ``from_array`` never produces (it yields the bare names), so its writer drops every note of a decoded
        array; here both spellings select the track.  Written with the built-in SMF writer instead of miditoolkit.
        Returns {track: [(velocity, pitch, start_tick, end_tick)]}, tempos, chords."""
        from . import smf
        npitch = len(DEFAULT_PITCH_RANGE)
        temp_notes, temp_chords, temp_tempos = [], [], []
        position, track = -1, ''
        ev = events
        for i in range(len(ev) - 3):
            nm = ev[i].name
            if nm == 'bar' and i > 0:
                temp_notes.append('bar'); temp_chords.append('bar'); temp_tempos.append('bar')
                track = ''
            elif nm == 'position':
                position = int(ev[i].value) - 1
            elif nm.startswith('track') or nm in DEFAULT_TRACKS:
                track = nm.split('_')[-1]
            elif nm == 'note_velocity' and ev[i + 1].name == 'note_on' and ev[i + 2].name == 'note_duration':
                velocity = int(DEFAULT_VELOCITY_BINS[int(ev[i].value)])
                v = int(ev[i + 1].value)
                if track == 'drum':
                    v = v + npitch if v < npitch else v
                    pitch = v + DEFAULT_DRUM_TYPE.start - npitch
                else:
                    v = v - npitch if v >= npitch else v
                    pitch = v + DEFAULT_PITCH_RANGE.start
                temp_notes.append([position, velocity, pitch, int(DEFAULT_DURATION_BINS[int(ev[i + 2].value)]), track])
            elif nm == 'chord':
                temp_chords.append([position, ev[i].value])
            elif nm == 'tempo_class' and ev[i + 1].name == 'tempo_value':
                position = int(ev[i].value)                      # as the reference does (MuMIDI.py:620)
                temp_tempos.append([position, DEFAULT_TEMPO_INTERVALS[ev[i].value].start + int(ev[i + 1].value)])
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

        notes = {}
        for st, vel, pitch, dur, trk in on_grid(temp_notes):
            notes.setdefault(trk, []).append((vel, pitch, st, st + dur))
        chords = on_grid(temp_chords)
        tempos = on_grid(temp_tempos)
        insts = [(instrument_numbers[t][0], t == 'drum', t, notes[t]) for t in DEFAULT_TRACKS if notes.get(t)]
        smf.write_ticks_multi(output_path, insts, DEFAULT_RESOLUTION, [(st, bpm) for st, bpm in tempos],
                              [(st, text) for st, text in chords])
        return notes, tempos, chords
