"""TEST INFRASTRUCTURE ONLY - CPU oracle of the sample-rate / channel conversion that precedes fbank.

The reference converts audio with an ffmpeg subprocess (`-ar 16000 -ac 1 -f wav -acodec pcm_s16le`,
speaker_detection_backends/audio_profiles.py:70-100; speechmatics_backend.py:231-281; speaker_samples:280-326).
ffmpeg is absent from this image and from the GPU box, so its resampler's output cannot be captured:
**parity unpinned** against ffmpeg.  What is pinned is this restatement: a Kaiser-windowed-sinc polyphase FIR
evaluated in INTEGER arithmetic (s16 samples x Q30 int32 taps, int64 accumulation, round-half-up shift, s16
saturation), so the GPU kernel must match it bit for bit.

    y[n] = sat16( ( sum_k taps[(n*M) mod L][k] * mono[ floor(n*M/L) + k - K/2 + 1 ] + 2^29 ) >> 30 )
    mono[i] = floor( (sum_c x[i][c] + C//2) / C ),    samples outside [0, n_in) are zero
    taps[p][k] = Q30( g(p/L - (k - K/2 + 1)) ),  g(t) = fc * sinc(fc t) * kaiser_beta(t / W),  fc = rolloff*min(1, L/M),
    W = zeros / min(1, L/M);  every phase is normalised to sum to exactly 2^30 (unit DC gain).
"""
from math import gcd

import numpy as np


def ratio(rate_in: int, rate_out: int):
    g = gcd(int(rate_in), int(rate_out))
    return int(rate_out) // g, int(rate_in) // g            # L (up), M (down)


def design_taps(rate_in: int, rate_out: int, zeros: int = 16, rolloff: float = 0.945, beta: float = 9.0):
    """-> (taps int32 [L, K], L, M, K)."""
    L, M = ratio(rate_in, rate_out)
    if L == 1 and M == 1:                                     # same rate: pure channel down-mix, no filtering
        return np.array([[1 << 30, 0]], dtype=np.int32), 1, 1, 2
    scale = min(1.0, L / M)
    W = zeros / scale                                         # half width in input samples
    K = 2 * int(np.ceil(W))
    fc = rolloff * scale
    p = np.arange(L, dtype=np.float64)[:, None] / L
    k = np.arange(K, dtype=np.float64)[None, :]
    t = p - (k - K // 2 + 1)
    win = np.where(np.abs(t) < W, np.i0(beta * np.sqrt(np.clip(1.0 - (t / W) ** 2, 0.0, None))) / np.i0(beta), 0.0)
    g = fc * np.sinc(fc * t) * win
    g = g / g.sum(axis=1, keepdims=True)
    q = np.floor(g * (1 << 30) + 0.5).astype(np.int64)
    # exact unit DC gain per phase: the residual goes to the largest tap
    resid = (1 << 30) - q.sum(axis=1)
    q[np.arange(L), np.argmax(q, axis=1)] += resid
    assert np.abs(q).max() < 2 ** 31
    return q.astype(np.int32), L, M, K


def out_len(n_in: int, L: int, M: int) -> int:
    return (int(n_in) * L + M - 1) // M


def downmix(x: np.ndarray) -> np.ndarray:
    """x int16 [n] or [n, C] -> mono int64 [n] (floor of the rounded mean)."""
    x = np.asarray(x)
    if x.ndim == 1:
        return x.astype(np.int64)
    C = x.shape[1]
    return (x.astype(np.int64).sum(axis=1) + C // 2) // C


def resample_s16(x: np.ndarray, taps: np.ndarray, L: int, M: int) -> np.ndarray:
    """x int16 [n] or [n, C]; taps int32 [L, K] -> int16 [ceil(n*L/M)]."""
    mono = downmix(x)
    n_in = len(mono)
    K = taps.shape[1]
    n_out = out_len(n_in, L, M)
    n = np.arange(n_out, dtype=np.int64)
    pos = n * M
    i0 = pos // L
    ph = pos - i0 * L
    pad = K
    xp = np.concatenate([np.zeros(pad, np.int64), mono, np.zeros(pad + 1, np.int64)])
    acc = np.zeros(n_out, dtype=np.int64)
    h = taps.astype(np.int64)
    base = i0 - (K // 2 - 1) + pad
    for k in range(K):
        acc += h[ph, k] * xp[base + k]
    y = (acc + (1 << 29)) >> 30
    return np.clip(y, -32768, 32767).astype(np.int16)


def to_s16(raw: np.ndarray, kind: str) -> np.ndarray:
    """Sample-format conversion to s16 (round half up on the dropped bits, saturate)."""
    if kind == "u8":
        return ((raw.astype(np.int32) - 128) << 8).astype(np.int16)
    if kind == "s16":
        return raw.astype(np.int16)
    if kind == "s24":
        return np.clip((raw.astype(np.int64) + 128) >> 8, -32768, 32767).astype(np.int16)
    if kind == "s32":
        return np.clip((raw.astype(np.int64) + 32768) >> 16, -32768, 32767).astype(np.int16)
    if kind in ("f32", "f64"):
        return np.clip(np.floor(raw.astype(np.float64) * 32768.0 + 0.5), -32768, 32767).astype(np.int16)
    raise ValueError(kind)
