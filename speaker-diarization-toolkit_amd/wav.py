"""RIFF/WAVE reader for the backend's input contract (16 kHz mono s16le PCM,
speaker_detection_backends/audio_profiles.py:25-29).  The reference converts and cuts audio with ffmpeg
subprocesses (speechmatics_backend.py:231-281; speaker_samples:280-326); here the file is read once, segments are
sliced from the sample array, and a WAVE file in another PCM layout (8/16/24/32-bit integer, 32/64-bit float, any
channel count, any sample rate) is converted on the GPU (`Engine.resample_s16`).  `read_wav_s16` is the strict
reader (contract format only); anything that is not RIFF/WAVE PCM is rejected with the ffmpeg command line that
produces the contract format (the toolkit's own conversion step)."""
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


def _chunks(path: Path, profile: AudioProfile):
    data = Path(path).read_bytes()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise AudioFormatError(f"{path}: not a RIFF/WAVE file; convert with: ffmpeg -i IN {' '.join(format_ffmpeg_args(profile))} OUT.wav")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None or len(fmt) < 16:
        raise AudioFormatError(f"{path}: missing fmt/data chunk")
    return fmt, pcm


def samples_to_s16(raw: np.ndarray, kind: str) -> np.ndarray:
    """Integer / float sample formats -> s16: keep the top 16 bits, round half up on the dropped ones, saturate."""
    if kind == "u8":
        return ((raw.astype(np.int32) - 128) << 8).astype(np.int16)
    if kind == "s16":
        return raw.astype(np.int16, copy=False)
    if kind == "s24":
        return np.clip((raw.astype(np.int64) + 128) >> 8, -32768, 32767).astype(np.int16)
    if kind == "s32":
        return np.clip((raw.astype(np.int64) + 32768) >> 16, -32768, 32767).astype(np.int16)
    if kind in ("f32", "f64"):
        return np.clip(np.floor(raw.astype(np.float64) * 32768.0 + 0.5), -32768, 32767).astype(np.int16)
    raise AudioFormatError(f"unsupported sample format {kind}")


def read_wav(path: Path, profile: Optional[AudioProfile] = None) -> Tuple[np.ndarray, int]:
    """Any PCM RIFF/WAVE -> (int16 [n, channels], sample_rate)."""
    profile = profile or AudioProfile()
    fmt, pcm = _chunks(Path(path), profile)
    tag, ch, rate, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:                      # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
        tag = struct.unpack("<H", fmt[24:26])[0]
    if ch < 1 or rate < 1:
        raise AudioFormatError(f"{path}: {ch} channels at {rate} Hz")
    if tag == 1 and bits in (8, 16, 24, 32):
        if bits == 8:
            raw, kind = np.frombuffer(pcm, dtype=np.uint8), "u8"
        elif bits == 16:
            raw, kind = np.frombuffer(pcm[:len(pcm) // 2 * 2], dtype="<i2"), "s16"
        elif bits == 24:
            b = np.frombuffer(pcm[:len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            raw, kind = ((b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)) ^ 0x800000) - 0x800000, "s24"
        else:
            raw, kind = np.frombuffer(pcm[:len(pcm) // 4 * 4], dtype="<i4"), "s32"
    elif tag == 3 and bits in (32, 64):
        raw, kind = np.frombuffer(pcm[:len(pcm) // (bits // 8) * (bits // 8)], dtype="<f4" if bits == 32 else "<f8"), f"f{bits}"
    else:
        raise AudioFormatError(
            f"{path}: WAVE format tag {tag} / {bits} bit is not linear PCM; convert with: ffmpeg -i IN {' '.join(format_ffmpeg_args(profile))} OUT.wav")
    s16 = samples_to_s16(raw, kind)
    n = len(s16) // ch
    return s16[:n * ch].reshape(n, ch), int(rate)


def decode_to_profile(path: Path, engine, profile: Optional[AudioProfile] = None) -> np.ndarray:
    """WAVE file -> mono int16 samples at the profile's rate.  Files already in the contract format are returned as
    read; anything else is down-mixed and resampled on the GPU (`sdk_resample_s16`, integer polyphase FIR)."""
    profile = profile or AudioProfile()
    x, rate = read_wav(path, profile)
    if rate == profile.sample_rate and x.shape[1] == profile.channels == 1:
        return np.ascontiguousarray(x[:, 0])
    if x.shape[0] == 0:
        return np.zeros((0,), dtype=np.int16)
    return engine.resample_s16_host(np.ascontiguousarray(x), rate, profile.sample_rate)      # ops.Engine and lite.LiteEngine both have it


def write_wav_s16(path: Path, samples: np.ndarray, rate: int = 16000) -> None:
    pcm = np.asarray(samples, dtype="<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16)
    Path(path).write_bytes(hdr + b"data" + struct.pack("<I", len(pcm)) + pcm)


def window_starts(n: int, segments: Optional[List[Tuple[float, float]]], rate: int = 16000,
                  window_s: float = 2.0, hop_s: float = 1.0, min_s: float = 0.5) -> Tuple[np.ndarray, List[Tuple[float, float]], int]:
    """The windows of cut_windows as a TABLE of first samples (int32 [B]) + the spans covered + the window length W: what crosses PCIe next
    to the recording itself when the windows are cut on the device (sdk_fbank_windows); a window that runs past sample n is zero padded there."""
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
            if pos - H + W < b:                      # this range's last window stops short of its end: add one flush with it
                starts.append(b - W)
    spans = [(a / rate, min(n, a + W) / rate) for a in starts]
    return np.asarray(starts, dtype=np.int32), spans, W


def materialise_windows(samples: np.ndarray, starts: np.ndarray, W: int) -> np.ndarray:
    """[B, W] int16 host copy of the windows of a start table, zero padded past the end of `samples` (tests, and hosts that keep the old form)."""
    out = np.zeros((len(starts), W), dtype=np.int16)
    for i, a in enumerate(starts.tolist()):
        chunk = samples[a:a + W]
        out[i, :len(chunk)] = chunk
    return out


def cut_windows(samples: np.ndarray, segments: Optional[List[Tuple[float, float]]], rate: int = 16000,
                window_s: float = 2.0, hop_s: float = 1.0, min_s: float = 0.5) -> Tuple[np.ndarray, List[Tuple[float, float]]]:
    """Slice (start, end) second ranges into fixed `window_s` windows (the unit the forward pass
    batches: 2 s = 32 000 samples = 201 frames).  Ranges shorter than the window are extended
    symmetrically (clamped to the file, then zero padded); longer ones are covered with hop `hop_s`.
    Returns (pcm [B, window] int16, [(start, end)] actually covered)."""
    starts, spans, W = window_starts(len(samples), segments, rate, window_s, hop_s, min_s)
    return materialise_windows(samples, starts, W), spans


BUCKETS_S = (0.5, 1.0, 1.5, 2.0)      # window lengths the forward is launched with: 51 / 101 / 151 / 201 frames


def range_starts(n: int, ranges: List[Tuple[float, float]], rate: int = 16000, buckets_s: Tuple[float, ...] = BUCKETS_S, hop_s: float = 1.0):
    """cut_ranges as start tables: (starts_by_len {window samples S: int32 [B_S]}, windows [(range index, S, row in the S table, start s, end s)],
    dropped [range index]).  Every window lies inside the recording (nothing is padded)."""
    sizes = sorted(int(round(b * rate)) for b in buckets_s)
    hop = int(round(hop_s * rate))
    starts_by = {S: [] for S in sizes}
    windows, dropped = [], []
    for ri, (s, e) in enumerate(ranges):
        a, b = max(0, int(round(s * rate))), min(n, int(round(e * rate)))
        fit = [S for S in sizes if S <= b - a]
        if not fit:
            dropped.append(ri)
            continue
        S = fit[-1]
        if S == sizes[-1]:
            st = list(range(a, b - S + 1, hop))
            if st[-1] + S < b:
                st.append(b - S)
        else:
            st = [a] if b - a == S else [a, b - S]
        for x in st:
            windows.append((ri, S, len(starts_by[S]), x / rate, (x + S) / rate))
            starts_by[S].append(x)
    return {S: np.asarray(st, dtype=np.int32) for S, st in starts_by.items() if st}, windows, dropped


def cut_ranges(samples: np.ndarray, ranges: List[Tuple[float, float]], rate: int = 16000, buckets_s: Tuple[float, ...] = BUCKETS_S,
               hop_s: float = 1.0):
    """True-length windows for (start, end) ranges that belong to ONE speaker each (sentences, enrollment segments).

    A window never leaves its range - nothing is widened into the neighbouring speech and nothing is zero padded:
      * a range at least as long as the largest bucket is covered by largest-bucket windows at `hop_s`, the last one flush
        with the range's end (the whole-recording rule, applied inside the range);
      * a shorter range takes the largest bucket that fits and is covered by one or two such windows (first flush with
        its start, second flush with its end);
      * a range shorter than the smallest bucket is dropped (returned in `dropped`).
    Returns (pcm_by_len {window samples: int16 [B, S]}, windows [(range index, S, row in pcm_by_len[S], start s, end s)],
    dropped [range index]).  The forward takes a uniform length per launch, so the caller runs one launch per bucket."""
    starts_by, windows, dropped = range_starts(len(samples), ranges, rate, buckets_s, hop_s)
    pcm_by_len = {S: np.stack([samples[x:x + S] for x in st.tolist()]).astype(np.int16) for S, st in starts_by.items()}
    return pcm_by_len, windows, dropped
