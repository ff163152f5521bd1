"""Minimal RIFF/WAVE reader for the backend's input contract (16 kHz mono s16le PCM,
speaker_detection_backends/audio_profiles.py:25-29).  The reference cuts segments with an ffmpeg
subprocess (speechmatics_backend.py:231-281); here the file is read once and segments are sliced
from the sample array.  Anything that is not already in the contract format is rejected with the
ffmpeg command line that produces it (the toolkit's own conversion step)."""
from __future__ import annotations

import struct
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np

from .audio_contract import AudioProfile, format_ffmpeg_args


class AudioFormatError(ValueError):
    pass


def read_wav_s16(path: Path, profile: Optional[AudioProfile] = None) -> np.ndarray:
    """Return the samples as int16 [n]."""
    profile = profile or AudioProfile()
    data = Path(path).read_bytes()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise AudioFormatError(f"{path}: not a RIFF/WAVE file; convert with: ffmpeg -i IN {' '.join(format_ffmpeg_args(profile))} OUT.wav")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise AudioFormatError(f"{path}: missing fmt/data chunk")
    tag, ch, rate, _, _, bits = fmt
    if tag not in (1, 0xFFFE) or ch != profile.channels or rate != profile.sample_rate or bits != profile.bit_depth:
        raise AudioFormatError(
            f"{path}: {rate} Hz / {ch} ch / {bits} bit (tag {tag}) does not match the backend's audio profile; "
            f"convert with: ffmpeg -i IN {' '.join(format_ffmpeg_args(profile))} OUT.wav")
    return np.frombuffer(pcm[:len(pcm) // 2 * 2], dtype="<i2").astype(np.int16, copy=False)


def write_wav_s16(path: Path, samples: np.ndarray, rate: int = 16000) -> None:
    pcm = np.asarray(samples, dtype="<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16)
    Path(path).write_bytes(hdr + b"data" + struct.pack("<I", len(pcm)) + pcm)


def cut_windows(samples: np.ndarray, segments: Optional[List[Tuple[float, float]]], rate: int = 16000,
                window_s: float = 2.0, hop_s: float = 1.0, min_s: float = 0.5) -> Tuple[np.ndarray, List[Tuple[float, float]]]:
    """Slice (start, end) second ranges into fixed `window_s` windows (the unit the forward pass
    batches: 2 s = 32 000 samples = 201 frames).  Ranges shorter than the window are extended
    symmetrically (clamped to the file, then zero padded); longer ones are covered with hop `hop_s`.
    Returns (pcm [B, window] int16, [(start, end)] actually covered)."""
    n = len(samples)
    W = int(round(window_s * rate))
    if segments is None:
        segments = [(0.0, n / rate)]
    starts: List[int] = []
    for s, e in segments:
        a, b = max(0, int(round(s * rate))), min(n, int(round(e * rate)))
        if b - a < int(min_s * rate):
            continue
        if b - a <= W:
            mid = (a + b) // 2
            starts.append(min(max(0, mid - W // 2), max(0, n - W)))
        else:
            H = int(round(hop_s * rate))
            pos = a
            while pos + W <= b:
                starts.append(pos)
                pos += H
            if starts and starts[-1] + W < b:
                starts.append(b - W)
    out = np.zeros((len(starts), W), dtype=np.int16)
    spans = []
    for i, a in enumerate(starts):
        chunk = samples[a:a + W]
        out[i, :len(chunk)] = chunk
        spans.append((a / rate, min(n, a + W) / rate))
    return out, spans
