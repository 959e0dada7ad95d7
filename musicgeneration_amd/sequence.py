"""MIDI-like event codec (array side), mirroring mg/model/utils/sequence.py (and its twin
mg/model/MusicTransformer/sequence.py): ``EventSeq.from_array / to_array / feat_dims / feat_ranges /
dim``, ``Event``, plus the note-level helpers that need no MIDI library.

Vocabulary (308 ids): note_on 0-87 (pitch 21..108), note_off 88-175, velocity 176-207 (32 bins),
time_shift 208-307 (0.01 s .. 1.00 s).  Index work is integer and bit-exact with the reference
(tests/test_codecs.py against tests/golden/g5_codecs.json).  MIDI file I/O uses pretty_midi when it
is installed (optional; absent in this image)."""
from __future__ import annotations

import copy

import numpy as np

from . import _vocab

DEFAULT_SAVING_PROGRAM = 1
DEFAULT_LOADING_PROGRAMS = range(128)
DEFAULT_RESOLUTION = 220
DEFAULT_TEMPO = 120
DEFAULT_VELOCITY = 64
DEFAULT_PITCH_RANGE = range(21, 109)
DEFAULT_VELOCITY_RANGE = range(21, 109)
DEFAULT_NORMALIZATION_BASELINE = 60

USE_VELOCITY = True
BEAT_LENGTH = 60 / DEFAULT_TEMPO
DEFAULT_TIME_SHIFT_BINS = 0.01 * np.arange(1, 101)
DEFAULT_VELOCITY_STEPS = 32
DEFAULT_NOTE_LENGTH = BEAT_LENGTH * 2
MIN_NOTE_LENGTH = BEAT_LENGTH / 2


class Note:
    """Minimal stand-in for pretty_midi.Note (velocity, pitch, start, end)."""

    def __init__(self, velocity, pitch, start, end):
        self.velocity, self.pitch, self.start, self.end = velocity, pitch, start, end

    def __repr__(self):
        return 'Note(start={}, end={}, pitch={}, velocity={})'.format(self.start, self.end, self.pitch, self.velocity)


class NoteSeq:
    """sequence.py:44-124 (the parts that do not need a MIDI parser)."""

    def __init__(self, notes=[]):
        self.notes = []
        if notes:
            self.add_notes(sorted(notes, key=lambda note: note.start))

    def add_notes(self, notes):
        self.notes += notes
        self.notes.sort(key=lambda note: note.start)

    @staticmethod
    def from_midi_file(path, *args, **kwargs):
        """sequence.py:44-46 -> from_midi (:37-41): non-drum notes of the selected programs.  Uses pretty_midi when it
        is installed (the reference's parser), otherwise the built-in SMF reader (smf.py)."""
        programs = kwargs.get('programs', DEFAULT_LOADING_PROGRAMS)
        try:
            from pretty_midi import PrettyMIDI
        except ImportError:
            from . import smf
            return NoteSeq([Note(v, p, s, e) for v, p, s, e in smf.read_notes(path, programs)])
        import itertools
        midi = PrettyMIDI(path)
        notes = itertools.chain(*[inst.notes for inst in midi.instruments
                                  if inst.program in programs and not inst.is_drum])
        return NoteSeq(list(notes))

    def to_midi_file(self, path, program=DEFAULT_SAVING_PROGRAM, resolution=DEFAULT_RESOLUTION,
                     tempo=DEFAULT_TEMPO):
        """sequence.py:65-77: one instrument, constant tempo.  pretty_midi when installed, else smf.write_notes."""
        try:
            from pretty_midi import PrettyMIDI, Instrument, Note as PMNote
        except ImportError:
            from . import smf
            smf.write_notes(path, [(n.velocity, n.pitch, n.start, n.end) for n in self.notes], program, resolution, tempo)
            return
        midi = PrettyMIDI(resolution=resolution, initial_tempo=tempo)
        inst = Instrument(program, False, 'NoteSeq')
        inst.notes = [PMNote(int(n.velocity), int(n.pitch), n.start, n.end) for n in self.notes]
        midi.instruments.append(inst)
        midi.write(path)


class Event:
    def __init__(self, type, time, value):
        self.type = type
        self.time = time
        self.value = value

    def __repr__(self):
        return 'Event(type={}, time={}, value={})'.format(self.type, self.time, self.value)


class EventSeq:
    pitch_range = DEFAULT_PITCH_RANGE
    velocity_range = DEFAULT_VELOCITY_RANGE
    velocity_steps = DEFAULT_VELOCITY_STEPS
    time_shift_bins = DEFAULT_TIME_SHIFT_BINS

    # ---- vocabulary ------------------------------------------------------------------------------
    @staticmethod
    def _layout():
        """(feature, slots) in id order -- sequence.py:204-212; the velocity feature exists only with USE_VELOCITY"""
        c = EventSeq
        keys = len(c.pitch_range)
        return (('note_on', keys), ('note_off', keys)) + ((('velocity', c.velocity_steps),) if USE_VELOCITY else ()) \
            + (('time_shift', len(c.time_shift_bins)),)

    @staticmethod
    def feat_dims():
        return _vocab.slots(EventSeq._layout())

    @staticmethod
    def feat_ranges():
        """sequence.py:214-221"""
        return _vocab.id_ranges(EventSeq._layout())

    @staticmethod
    def dim():
        return sum(n for _, n in EventSeq._layout())

    @staticmethod
    def get_velocity_bins():
        n = EventSeq.velocity_range.stop - EventSeq.velocity_range.start
        return np.arange(EventSeq.velocity_range.start, EventSeq.velocity_range.stop,
                         n / (EventSeq.velocity_steps - 1))

    # ---- index array <-> events --------------------------------------------------------------------
    @staticmethod
    def decode_arrays(event_indeces):
        """Vectorised core of from_array: -> (type_id int[T], value int[T], time float64[T]).
        type_id indexes feat_ranges() order; ids outside the vocabulary get type_id -1 (the
        reference silently drops them, sequence.py:189-196)."""
        idx = np.asarray(event_indeces).astype(np.int64).reshape(-1)
        starts = np.array([r.start for r in EventSeq.feat_ranges().values()] + [EventSeq.dim()], dtype=np.int64)
        tid = np.searchsorted(starts, idx, side='right') - 1
        valid = (idx >= 0) & (idx < EventSeq.dim())
        tid = np.where(valid, tid, -1)
        val = idx - starts[np.clip(tid, 0, len(starts) - 2)]
        names = list(EventSeq.feat_ranges().keys())
        ts = names.index('time_shift')
        # the reference accumulates time with sequential float adds: reproduce that order exactly
        inc = np.where(tid == ts, EventSeq.time_shift_bins[np.clip(val, 0, len(EventSeq.time_shift_bins) - 1)], 0.0)
        time = np.zeros(len(idx), dtype=np.float64)
        acc = 0.0
        for i in range(len(idx)):          # sequential: float addition is not associative
            time[i] = acc
            acc += inc[i]
        return tid, val, time

    @staticmethod
    def from_array(event_indeces):
        """sequence.py:185-198"""
        tid, val, time = EventSeq.decode_arrays(event_indeces)
        names = list(EventSeq.feat_ranges().keys())
        events = [Event(names[t], tm, int(v)) for t, v, tm in zip(tid, val, time) if t >= 0]
        return EventSeq(events)

    def __init__(self, events=()):
        """own copies of the events; an event's time is re-derived from the time_shift events before it
        (sequence.py:222-233) by sequential accumulation, starting from integer 0 like the reference"""
        source = list(events)
        if not all(isinstance(ev, Event) for ev in source):
            raise AssertionError("EventSeq takes Event objects")
        clock = 0
        self.events = []
        for ev in source:
            self.events.append(Event(ev.type, clock, copy.deepcopy(ev.value)))
            if ev.type == 'time_shift':
                clock = clock + EventSeq.time_shift_bins[ev.value]

    def to_array(self):
        """sequence.py:283-287"""
        feat_idxs = EventSeq.feat_ranges()
        idxs = [feat_idxs[event.type][event.value] for event in self.events]
        dtype = np.uint8 if EventSeq.dim() <= 256 else np.uint16
        return np.array(idxs, dtype=dtype)

    # ---- notes <-> events (no MIDI library needed) ---------------------------------------------------
    @staticmethod
    def from_note_seq(note_seq):
        """notes -> events (sequence.py:145-183): per in-range note a velocity bin (at the onset), note_on and note_off;
        stable sort by time; between consecutive events the gap is covered greedily by the largest time_shift bins."""
        vbins = EventSeq.get_velocity_bins()
        v_lo, v_hi = EventSeq.velocity_range.start, EventSeq.velocity_range.stop - 1
        first_pitch = EventSeq.pitch_range.start
        timeline = []                                    # (time, type, value) in emission order
        for n in note_seq.notes:
            if n.pitch not in EventSeq.pitch_range:
                continue
            key = n.pitch - first_pitch
            if USE_VELOCITY:
                timeline.append((n.start, 'velocity', int(np.searchsorted(vbins, min(max(n.velocity, v_lo), v_hi)))))
            timeline.append((n.start, 'note_on', key))
            timeline.append((n.end, 'note_off', key))
        timeline.sort(key=lambda item: item[0])          # stable, like the reference's list.sort on Event.time
        bins = EventSeq.time_shift_bins
        out = []
        for pos, (t, kind, value) in enumerate(timeline):
            out.append(Event(kind, t, value))
            if pos + 1 == len(timeline):
                break
            gap, used = timeline[pos + 1][0] - t, 0
            while gap - used >= bins[0]:
                j = int(np.searchsorted(bins, gap - used, side='right')) - 1
                out.append(Event('time_shift', t + used, j))
                used += bins[j]
        return EventSeq(out)

    def to_note_seq(self):
        """events -> notes (sequence.py:235-272): a clock advanced by time_shift, the current velocity bin, and the notes
        still sounding per pitch; a note_off closes its pitch's note (at least MIN_NOTE_LENGTH long), notes never closed
        last DEFAULT_NOTE_LENGTH."""
        vbins = EventSeq.get_velocity_bins()
        first_pitch = EventSeq.pitch_range.start
        clock, velocity = 0, DEFAULT_VELOCITY
        sounding = {}
        notes = []
        for ev in self.events:
            kind = ev.type
            if kind == 'time_shift':
                clock += EventSeq.time_shift_bins[ev.value]
            elif kind == 'velocity':
                velocity = vbins[min(ev.value, vbins.size - 1)]
            elif kind == 'note_on':
                note = Note(velocity, ev.value + first_pitch, clock, None)
                notes.append(note)
                sounding[note.pitch] = note
            elif kind == 'note_off':
                note = sounding.pop(ev.value + first_pitch, None)
                if note is not None:
                    note.end = max(clock, note.start + MIN_NOTE_LENGTH)
        for note in notes:
            if note.end is None:
                note.end = note.start + DEFAULT_NOTE_LENGTH
            note.velocity = int(note.velocity)
        return NoteSeq(notes)
