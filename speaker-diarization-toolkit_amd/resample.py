"""Host side of the audio conversion to the backend's AudioProfile (16 kHz mono s16,
speaker_detection_backends/audio_profiles.py:25-29): filter design and sizes for `sdk_resample_s16`.

The reference shells out to ffmpeg for this (`format_ffmpeg_args`, audio_profiles.py:70-100;
speechmatics_backend.py:231-281).  Here a RIFF/WAVE file of any PCM layout is converted on the GPU: rational
polyphase FIR, Kaiser-windowed sinc, integer arithmetic (s16 x Q30 taps, int64 accumulate), so the result is
reproducible bit for bit on any device.  Tap tables are designed once per (rate_in, rate_out) in float64 and
quantised here; the kernel only consumes the table."""
from __future__ import annotations

from functools import lru_cache
from math import gcd
from typing import Tuple

import numpy as np

ZEROS = 16          # sinc zero crossings kept on each side (at the narrower of the two Nyquist rates)
ROLLOFF = 0.945     # cut-off as a fraction of that Nyquist rate
BETA = 9.0          # Kaiser window shape (stop band about -90 dB)


def ratio(rate_in: int, rate_out: int) -> Tuple[int, int]:
    g = gcd(int(rate_in), int(rate_out))
    return int(rate_out) // g, int(rate_in) // g            # L (interpolation), M (decimation)


@lru_cache(maxsize=16)
def design_taps(rate_in: int, rate_out: int) -> Tuple[np.ndarray, int, int, int]:
    """-> (taps int32 [L, K] with every phase summing to exactly 2^30, L, M, K)."""
    L, M = ratio(rate_in, rate_out)
    if L == 1 and M == 1:                                     # same rate: pure channel down-mix, no filtering
        taps = np.array([[1 << 30, 0]], dtype=np.int32)
        taps.setflags(write=False)
        return taps, 1, 1, 2
    scale = min(1.0, L / M)
    W = ZEROS / scale
    K = 2 * int(np.ceil(W))
    fc = ROLLOFF * scale
    frac = np.arange(L, dtype=np.float64)[:, None] / L
    tap = np.arange(K, dtype=np.float64)[None, :]
    t = frac - (tap - K // 2 + 1)                              # time of input sample i0 + k - K/2 + 1 relative to the output
    inside = np.clip(1.0 - (t / W) ** 2, 0.0, None)
    win = np.where(np.abs(t) < W, np.i0(BETA * np.sqrt(inside)) / np.i0(BETA), 0.0)
    g = fc * np.sinc(fc * t) * win
    g = g / g.sum(axis=1, keepdims=True)
    q = np.floor(g * (1 << 30) + 0.5).astype(np.int64)
    q[np.arange(L), np.argmax(q, axis=1)] += (1 << 30) - q.sum(axis=1)
    if np.abs(q).max() >= 2 ** 31:
        raise ValueError(f"resample {rate_in}->{rate_out}: tap overflow")
    taps = q.astype(np.int32)
    taps.setflags(write=False)
    return taps, L, M, K


def out_len(n_in: int, L: int, M: int) -> int:
    return (int(n_in) * L + M - 1) // M
