"""k1 oracle: 80-bin log-mel filterbank features (CPU, float64 internally).

TEST INFRASTRUCTURE - see oracle/__init__.py.  **Parity unpinned**: the reference has
no feature extractor (speaker_detection_backends/audio_profiles.py:12-47 only fixes the
INPUT contract: 16 kHz, mono, s16le).  This restates the published recipe the ECAPA-TDNN
papers use (25 ms Hamming window, 10 ms hop, 400-point DFT, 80 triangular mel filters
0-8000 Hz, dB log, 80 dB dynamic-range floor, per-utterance mean normalisation):

    x[n]      = pcm[n] / 32768
    frame t   = x[t*160 - 200 : t*160 + 200]   (zero padded at both ends),  T = 1 + S // 160
    w[n]      = 0.54 - 0.46 cos(2 pi n / 400)                       (periodic Hamming)
    P[t,f]    = |sum_n w[n] frame_t[n] e^{-2 pi i f n / 400}|^2,   f = 0..200
    M[t,m]    = sum_f P[t,f] * melW[f,m]
    L[t,m]    = 10 log10(max(M[t,m], 1e-10));  L = max(L, max_{t,m} L - 80)
    out[t,m]  = L[t,m] - mean_t L[t,m]

Filter shape - a BUILD CHOICE, parity unpinned: melW uses the HTK/Slaney-style triangle whose three corners are consecutive
points of a grid equally spaced on the HTK mel scale (rise from point m to m+1, fall to m+2: the two sides have different
widths in Hz, and neighbouring filters sum to one between the outer centres).  Toolkits differ here (some use one band
width per filter, i.e. symmetric triangles in Hz); none is importable in this image and the reference pins none, so no
published recipe is claimed for the triangle shape.  tests/test_oracle_cpu.py checks this matrix against a second,
per-bin construction of the same definition.
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
N_MELS = 80
N_BINS = N_FFT // 2 + 1
AMIN = 1e-10
TOP_DB = 80.0


def num_frames(n_samples: int) -> int:
    return 1 + n_samples // HOP


def hamming_window() -> np.ndarray:
    n = np.arange(N_FFT, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * n / N_FFT)


def mel_matrix() -> np.ndarray:
    """[201, 80] triangular filters, HTK mel scale, unit peak, float64."""
    def hz2mel(f):
        return 2595.0 * np.log10(1.0 + f / 700.0)

    def mel2hz(m):
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)

    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(SAMPLE_RATE / 2.0), N_MELS + 2))
    freqs = np.linspace(0.0, SAMPLE_RATE / 2.0, N_BINS)
    W = np.zeros((N_BINS, N_MELS), dtype=np.float64)
    for m in range(N_MELS):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        up = (freqs - lo) / (ce - lo)
        down = (hi - freqs) / (hi - ce)
        W[:, m] = np.maximum(0.0, np.minimum(up, down))
    return W


def dft_matrices():
    """cos/sin DFT matrices with the window folded in: [400, 201] each, float64."""
    n = np.arange(N_FFT, dtype=np.float64)[:, None]
    f = np.arange(N_BINS, dtype=np.float64)[None, :]
    ang = 2.0 * np.pi * ((n * f) % N_FFT) / N_FFT
    w = hamming_window()[:, None]
    return w * np.cos(ang), -w * np.sin(ang)


def frames_of(pcm: np.ndarray) -> np.ndarray:
    """[B, S] int16 -> [B, T, 400] float64 centred, zero-padded frames."""
    pcm = np.atleast_2d(pcm)
    B, S = pcm.shape
    T = num_frames(S)
    x = np.zeros((B, S + N_FFT), dtype=np.float64)
    x[:, N_FFT // 2: N_FFT // 2 + S] = pcm.astype(np.float64) / 32768.0
    idx = (np.arange(T) * HOP)[:, None] + np.arange(N_FFT)[None, :]
    return x[:, idx]


def _round_significand(x: np.ndarray, bits: int) -> np.ndarray:
    """float64 -> nearest value with `bits` significand bits (error-budget switch only)."""
    m, e = np.frexp(x)
    return np.ldexp(np.round(m * (1 << bits)) / (1 << bits), e)


def power_spectrum(pcm: np.ndarray, dft_bits: int | None = None) -> np.ndarray:
    """dft_bits (error budget): round the windowed DFT matrices to that many significand bits - 16 models
    the GPU kernel's bf16 hi+lo split of the table (the int16 samples split exactly); None = the exact float64 table."""
    C, Sn = dft_matrices()
    if dft_bits is not None:
        C, Sn = _round_significand(C, dft_bits), _round_significand(Sn, dft_bits)
    fr = frames_of(pcm)
    re = fr @ C
    im = fr @ Sn
    return re * re + im * im


def fbank(pcm: np.ndarray, dft_bits: int | None = None) -> np.ndarray:
    """[B, S] int16 -> [B, T, 80] float32 mean-normalised log-mel features."""
    P = power_spectrum(pcm, dft_bits)
    M = P @ mel_matrix()
    L = 10.0 * np.log10(np.maximum(M, AMIN))
    peak = L.reshape(L.shape[0], -1).max(axis=1)[:, None, None]
    L = np.maximum(L, peak - TOP_DB)
    L = L - L.mean(axis=1, keepdims=True)
    return L.astype(np.float32)
