"""Dependency-free Standard MIDI File (SMF) reader / writer for the note level of the MIDI-like codec.

The reference goes through ``pretty_midi`` (sequence.py:37-41 ``NoteSeq.from_midi``, :65-73 ``to_midi``), which is
not installed here and not pinned by the reference either ("parity unpinned" for MIDI-file I/O).  This
module restates the published SMF 1.0 format for exactly what those two calls need:

* ``write_notes``: format-1 file, ``resolution`` ticks per quarter, one conductor track (tempo, 4/4) and one
  instrument track (program change, note-on / note-off with velocity 0 release) -- the layout pretty_midi writes.
* ``read_notes``: every track, running status, tempo map -> seconds, program per channel, drum channel 9 skipped;
  returns (velocity, pitch, start, end) tuples of the instruments whose program is in ``programs``.

It is host-side file I/O: it is never on the GPU path and has no kernel counterpart."""
from __future__ import annotations

import struct
from typing import Iterable, List, Sequence, Tuple

NoteTuple = Tuple[int, int, float, float]          # velocity, pitch, start [s], end [s]


def _vlq(n: int) -> bytes:
    """variable-length quantity (7 bits per byte, MSB = continuation)"""
    if n < 0:
        raise ValueError("negative delta time")
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append(0x80 | (n & 0x7F))
        n >>= 7
    return bytes(reversed(out))


def _chunk(tag: bytes, body: bytes) -> bytes:
    return tag + struct.pack(">I", len(body)) + body


def write_notes(path: str, notes: Sequence[NoteTuple], program: int = 1, resolution: int = 220, tempo: float = 120.0,
                is_drum: bool = False, name: str = "NoteSeq") -> None:
    """Write ``notes`` (seconds) as one instrument at a constant ``tempo`` (BPM)."""
    ticks_per_second = resolution * tempo / 60.0
    us_per_beat = int(round(6e7 / tempo))
    conductor = (b"\x00\xFF\x51\x03" + struct.pack(">I", us_per_beat)[1:]         # set tempo
                 + b"\x00\xFF\x58\x04\x04\x02\x18\x08"                            # 4/4
                 + b"\x00\xFF\x2F\x00")
    chan = 9 if is_drum else 0
    events: List[Tuple[int, int, bytes]] = []                                     # (tick, order, bytes)
    for vel, pitch, start, end in notes:
        vel, pitch = max(1, min(127, int(vel))), max(0, min(127, int(pitch)))
        t0, t1 = int(round(start * ticks_per_second)), int(round(end * ticks_per_second))
        t1 = max(t1, t0)
        events.append((t0, 1, bytes([0x90 | chan, pitch, vel])))
        events.append((t1, 0, bytes([0x90 | chan, pitch, 0])))                    # release = note-on with velocity 0
    events.sort(key=lambda e: (e[0], e[1]))                                       # releases before attacks on a tie
    nm = name.encode("ascii", "replace")
    body = bytearray(b"\x00\xFF\x03" + _vlq(len(nm)) + nm)
    body += b"\x00" + bytes([0xC0 | chan, int(program) & 0x7F])
    last = 0
    for tick, _, msg in events:
        body += _vlq(tick - last) + msg
        last = tick
    body += b"\x00\xFF\x2F\x00"
    with open(path, "wb") as f:
        f.write(_chunk(b"MThd", struct.pack(">HHH", 1, 2, resolution)))
        f.write(_chunk(b"MTrk", conductor))
        f.write(_chunk(b"MTrk", bytes(body)))


def write_ticks_multi(path: str, instruments, resolution: int = 480, tempo_changes: Sequence[Tuple[int, float]] = (),
                      markers: Sequence[Tuple[int, str]] = ()) -> None:
    """Tick-level writer (what miditoolkit's ``MidiFile.dump`` produces for REMI / MuMIDI ``write_midi``):
    ``instruments`` = [(program, is_drum, name, notes)], notes = (velocity, pitch, start_tick, end_tick);
    ``tempo_changes`` = (tick, bpm); ``markers`` = (tick, text) -- chord symbols go there.  Conductor track: tempo
    map, 4/4, markers; one track and one channel per instrument (drums on channel 9)."""
    cond: List[Tuple[int, int, bytes]] = [(0, 0, b"\xFF\x58\x04\x04\x02\x18\x08")]
    tc = sorted((int(t), float(b)) for t, b in tempo_changes) or [(0, 120.0)]
    if tc[0][0] != 0:
        tc.insert(0, (0, 120.0))
    for t, bpm in tc:
        cond.append((t, 1, b"\xFF\x51\x03" + struct.pack(">I", int(round(6e7 / max(bpm, 1e-3))))[1:]))
    for t, text in markers:
        tb = str(text).encode("ascii", "replace")
        cond.append((int(t), 2, b"\xFF\x06" + _vlq(len(tb)) + tb))
    cond.sort(key=lambda e: (e[0], e[1]))
    cbody, last = bytearray(), 0
    for tick, _, msg in cond:
        cbody += _vlq(tick - last) + msg
        last = tick
    cbody += b"\x00\xFF\x2F\x00"
    tracks, next_chan = [], 0
    for program, is_drum, name, notes in instruments:
        if is_drum:
            chan = 9
        else:
            chan = next_chan
            next_chan += 2 if next_chan == 8 else 1          # skip the percussion channel
            if chan > 15:
                raise ValueError("more than 15 melodic instruments")
        ev: List[Tuple[int, int, bytes]] = []
        for vel, pitch, t0, t1 in notes:
            vel, pitch = max(1, min(127, int(vel))), max(0, min(127, int(pitch)))
            t0, t1 = int(t0), max(int(t1), int(t0))
            ev.append((t0, 1, bytes([0x90 | chan, pitch, vel])))
            ev.append((t1, 0, bytes([0x90 | chan, pitch, 0])))
        ev.sort(key=lambda e: (e[0], e[1]))
        nm = str(name).encode("ascii", "replace")
        body = bytearray((b"\x00\xFF\x03" + _vlq(len(nm)) + nm) if nm else b"")
        body += b"\x00" + bytes([0xC0 | chan, int(program) & 0x7F])
        last = 0
        for tick, _, msg in ev:
            body += _vlq(tick - last) + msg
            last = tick
        body += b"\x00\xFF\x2F\x00"
        tracks.append(bytes(body))
    with open(path, "wb") as f:
        f.write(_chunk(b"MThd", struct.pack(">HHH", 1, 1 + len(tracks), resolution)))
        f.write(_chunk(b"MTrk", bytes(cbody)))
        for body in tracks:
            f.write(_chunk(b"MTrk", body))


def write_ticks(path: str, notes: Sequence[Tuple[int, int, int, int]], resolution: int = 480,
                tempo_changes: Sequence[Tuple[int, float]] = (), markers: Sequence[Tuple[int, str]] = (),
                program: int = 0, is_drum: bool = False) -> None:
    """single-instrument form of ``write_ticks_multi`` (REMI.write_midi, utils/REMI.py:651-670)"""
    write_ticks_multi(path, [(program, is_drum, "", notes)], resolution, tempo_changes, markers)


def _read_vlq(buf: bytes, i: int) -> Tuple[int, int]:
    n = 0
    while True:
        b = buf[i]
        i += 1
        n = (n << 7) | (b & 0x7F)
        if not b & 0x80:
            return n, i


def _parse_track(buf: bytes):
    """-> list of (abs_tick, kind, data): kind in {'tempo','on','off','program'}"""
    out = []
    i, tick, status = 0, 0, 0
    n = len(buf)
    while i < n:
        d, i = _read_vlq(buf, i)
        tick += d
        b = buf[i]
        if b == 0xFF:                                    # meta
            typ = buf[i + 1]
            ln, j = _read_vlq(buf, i + 2)
            data = buf[j:j + ln]
            i = j + ln
            if typ == 0x51 and ln == 3:
                out.append((tick, "tempo", int.from_bytes(data, "big")))
            if typ == 0x2F:
                break
            continue
        if b in (0xF0, 0xF7):                            # sysex
            ln, j = _read_vlq(buf, i + 1)
            i = j + ln
            continue
        if b & 0x80:
            status = b
            i += 1
        hi, ch = status & 0xF0, status & 0x0F            # running status otherwise
        if hi in (0x80, 0x90, 0xA0, 0xB0, 0xE0):
            p1, p2 = buf[i], buf[i + 1]
            i += 2
            if hi == 0x90 and p2 > 0:
                out.append((tick, "on", (ch, p1, p2)))
            elif hi == 0x80 or hi == 0x90:
                out.append((tick, "off", (ch, p1)))
        elif hi in (0xC0, 0xD0):
            p1 = buf[i]
            i += 1
            if hi == 0xC0:
                out.append((tick, "program", (ch, p1)))
        else:
            raise ValueError(f"unsupported MIDI status byte 0x{status:02x}")
    return out


def read_notes(path: str, programs: Iterable[int] = range(128)) -> List[NoteTuple]:
    """All non-drum notes of instruments whose program is in ``programs``, times in seconds."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] != b"MThd":
        raise ValueError("not a Standard MIDI File")
    hlen = struct.unpack(">I", raw[4:8])[0]
    fmt, ntrk, division = struct.unpack(">HHH", raw[8:14])
    if division & 0x8000:
        raise ValueError("SMPTE time division is not supported")
    i = 8 + hlen
    tracks = []
    while i + 8 <= len(raw) and len(tracks) < ntrk:
        tag, ln = raw[i:i + 4], struct.unpack(">I", raw[i + 4:i + 8])[0]
        if tag == b"MTrk":
            tracks.append(_parse_track(raw[i + 8:i + 8 + ln]))
        i += 8 + ln
    # tempo map (format 1: tempo events apply to every track)
    tempi = sorted((t, v) for tr in tracks for (t, k, v) in tr if k == "tempo")
    if not tempi or tempi[0][0] != 0:
        tempi.insert(0, (0, 500000))
    seg = []                                             # (tick0, seconds0, seconds_per_tick)
    sec = 0.0
    for k, (t, us) in enumerate(tempi):
        if k:
            sec += (t - tempi[k - 1][0]) * seg[-1][2]
        seg.append((t, sec, us * 1e-6 / division))

    def to_sec(tick):
        lo, hi = 0, len(seg) - 1
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if seg[mid][0] <= tick:
                lo = mid
            else:
                hi = mid - 1
        t0, s0, spt = seg[lo]
        return s0 + (tick - t0) * spt

    allowed = set(programs)
    notes: List[NoteTuple] = []
    for tr in tracks:
        prog = {}
        open_notes = {}
        for tick, kind, v in sorted(tr, key=lambda e: (e[0], {"program": 0, "tempo": 0, "off": 1, "on": 2}[e[1]])):
            if kind == "program":
                prog[v[0]] = v[1]
            elif kind == "on":
                open_notes.setdefault((v[0], v[1]), []).append((tick, v[2]))
            elif kind == "off":
                lst = open_notes.get((v[0], v[1]))
                if lst:
                    t0, vel = lst.pop(0)
                    ch = v[0]
                    if ch != 9 and prog.get(ch, 0) in allowed:
                        notes.append((vel, v[1], to_sec(t0), to_sec(tick)))
    notes.sort(key=lambda nt: nt[2])
    return notes


def read_ticks(path: str):
    """-> dict(resolution, notes=[(velocity, pitch, start_tick, end_tick)], tempo_changes=[(tick, bpm)], markers=[(tick, text)])
    of every non-drum channel (tick-level twin of ``read_notes``; used to round-trip ``write_ticks`` in the tests)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] != b"MThd":
        raise ValueError("not a Standard MIDI File")
    hlen = struct.unpack(">I", raw[4:8])[0]
    _, ntrk, division = struct.unpack(">HHH", raw[8:14])
    i, notes, tempi, markers = 8 + hlen, [], [], []
    by_channel, programs = {}, {}
    for _ in range(ntrk):
        tag, ln = raw[i:i + 4], struct.unpack(">I", raw[i + 4:i + 8])[0]
        buf = raw[i + 8:i + 8 + ln]
        i += 8 + ln
        if tag != b"MTrk":
            continue
        # markers need the raw meta events: a light second parse
        j, tick, status, open_notes = 0, 0, 0, {}
        while j < len(buf):
            d, j = _read_vlq(buf, j)
            tick += d
            b = buf[j]
            if b == 0xFF:
                typ = buf[j + 1]
                n, k = _read_vlq(buf, j + 2)
                data = buf[k:k + n]
                j = k + n
                if typ == 0x51:
                    tempi.append((tick, 6e7 / int.from_bytes(data, "big")))
                elif typ == 0x06:
                    markers.append((tick, data.decode("ascii", "replace")))
                elif typ == 0x2F:
                    break
                continue
            if b in (0xF0, 0xF7):
                n, k = _read_vlq(buf, j + 1)
                j = k + n
                continue
            if b & 0x80:
                status = b
                j += 1
            hi, ch = status & 0xF0, status & 0x0F
            if hi in (0x80, 0x90, 0xA0, 0xB0, 0xE0):
                p1, p2 = buf[j], buf[j + 1]
                j += 2
                if hi == 0x90 and p2 > 0:
                    open_notes.setdefault((ch, p1), []).append((tick, p2))
                elif hi in (0x80, 0x90) and open_notes.get((ch, p1)):
                    t0, vel = open_notes[(ch, p1)].pop(0)
                    by_channel.setdefault(ch, []).append((vel, p1, t0, tick))
                    if ch != 9:
                        notes.append((vel, p1, t0, tick))
            elif hi in (0xC0, 0xD0):
                if hi == 0xC0:
                    programs[ch] = buf[j]
                j += 1
    notes.sort(key=lambda n: (n[2], n[1]))
    for v in by_channel.values():
        v.sort(key=lambda n: (n[2], n[1]))
    return {"resolution": division, "notes": notes, "tempo_changes": sorted(tempi), "markers": sorted(markers),
            "by_channel": by_channel, "programs": programs}

