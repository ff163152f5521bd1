"""Audio conversion to the AudioProfile, CPU side: the oracle's own properties, the host filter design
against the oracle's restatement, the sample-format rules and the WAVE reader.  No GPU."""
import struct

import numpy as np
import pytest

from conftest import sub
from oracle import resample as ors

R = sub("resample")
WAV = sub("wav")

RATES = (8000, 11025, 22050, 32000, 44100, 48000, 96000)


@pytest.mark.parametrize("rate", RATES + (16000,))
def test_host_design_equals_oracle_design(rate):
    t, L, M, K = R.design_taps(rate, 16000)
    to, Lo, Mo, Ko = ors.design_taps(rate, 16000)
    assert (L, M, K) == (Lo, Mo, Ko) and t.dtype == np.int32 and np.array_equal(t, to)
    assert (t.astype(np.int64).sum(axis=1) == 1 << 30).all()          # unit DC gain, every phase
    assert R.out_len(12345, L, M) == ors.out_len(12345, L, M) == -(-12345 * L // M)


@pytest.mark.parametrize("rate", RATES)
def test_oracle_constant_tone_and_alias(rate):
    t, L, M, K = ors.design_taps(rate, 16000)
    n = rate // 2
    edge = 4 * K
    y = ors.resample_s16(np.full(n, -4321, np.int16), t, L, M)
    assert len(y) == ors.out_len(n, L, M) and set(y[edge:-edge].tolist()) == {-4321}
    # a 1 kHz tone (inside every pass band) comes out as the same tone at 16 kHz, within 2 LSB
    x = np.round(10000 * np.sin(2 * np.pi * 1000 * np.arange(n) / rate)).astype(np.int16)
    y = ors.resample_s16(x, t, L, M)
    ref = 10000 * np.sin(2 * np.pi * 1000 * np.arange(len(y)) / 16000)
    assert np.abs(y[edge:-edge] - ref[edge:-edge]).max() <= 2.0
    # a tone above the new Nyquist rate must not fold back (stop band <= 2 LSB at amplitude 10000: -74 dB)
    if rate > 16000:
        f = 0.5 * (8000 + rate / 2)
        x = np.round(10000 * np.sin(2 * np.pi * f * np.arange(n) / rate)).astype(np.int16)
        assert np.abs(ors.resample_s16(x, t, L, M)[edge:-edge]).max() <= 2


def test_oracle_identity_rate_is_identity():
    t, L, M, K = ors.design_taps(16000, 16000)
    x = np.random.default_rng(0).integers(-32768, 32768, 3000).astype(np.int16)
    assert (L, M) == (1, 1) and np.array_equal(ors.resample_s16(x, t, L, M), x)      # sinc sampled at integers = delta


def test_downmix_rounding():
    x = np.array([[1, 2], [-1, -2], [32767, 32767], [-32768, -32768], [3, -4], [5, 0]], dtype=np.int16)
    assert ors.downmix(x).tolist() == [2, -1, 32767, -32768, 0, 3]                   # floor((a + b + 1) / 2)
    x3 = np.array([[1, 1, 2], [-1, -1, -2]], dtype=np.int16)
    assert ors.downmix(x3).tolist() == [1, -1]                                       # floor((s + 1) / 3)


def test_sample_format_rules_match_oracle():
    rng = np.random.default_rng(1)
    cases = {"u8": rng.integers(0, 256, 500).astype(np.uint8), "s16": rng.integers(-32768, 32768, 500).astype(np.int16),
             "s24": np.concatenate([rng.integers(-(1 << 23), 1 << 23, 500), [8388607, -8388608, 127, 128, -128, -129]]).astype(np.int32),
             "s32": np.concatenate([rng.integers(-(1 << 31), 1 << 31, 500), [2147483647, -2147483648, 32767, 32768]]).astype(np.int32),
             "f32": np.concatenate([rng.uniform(-1.2, 1.2, 500), [1.0, -1.0, 0.5 / 32768, -0.5 / 32768]]).astype(np.float32)}
    for kind, raw in cases.items():
        assert np.array_equal(WAV.samples_to_s16(raw, kind), ors.to_s16(raw, kind)), kind
    assert WAV.samples_to_s16(np.array([8388607], np.int32), "s24")[0] == 32767           # saturates instead of wrapping
    assert WAV.samples_to_s16(np.array([1.0, -1.0], np.float32), "f32").tolist() == [32767, -32768]


def _wav_bytes(tag, ch, rate, bits, payload, extensible=False):
    if extensible:
        fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits, 22, bits, 0, tag) + b"\x00" * 14
    else:
        fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits)
    body = b"WAVEfmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd" + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_read_wav_layouts(tmp_path):
    rng = np.random.default_rng(2)
    s = rng.integers(-30000, 30000, (100, 2)).astype(np.int16)
    p = tmp_path / "a.wav"
    p.write_bytes(_wav_bytes(1, 2, 44100, 16, s.astype("<i2").tobytes()))
    x, rate = WAV.read_wav(p)
    assert rate == 44100 and x.shape == (100, 2) and np.array_equal(x, s)
    # 24-bit mono, WAVE_FORMAT_EXTENSIBLE header
    v = rng.integers(-(1 << 23), 1 << 23, 50).astype(np.int32)
    b = bytearray()
    for q in v.tolist():
        b += int(q & 0xFFFFFF).to_bytes(3, "little")
    p.write_bytes(_wav_bytes(1, 1, 48000, 24, bytes(b), extensible=True))
    x, rate = WAV.read_wav(p)
    assert rate == 48000 and np.array_equal(x[:, 0], ors.to_s16(v, "s24"))
    # float32 stereo, 8-bit mono
    f = rng.uniform(-1, 1, (40, 2)).astype("<f4")
    p.write_bytes(_wav_bytes(3, 2, 22050, 32, f.tobytes()))
    x, rate = WAV.read_wav(p)
    assert rate == 22050 and np.array_equal(x, ors.to_s16(f.reshape(-1), "f32").reshape(40, 2))
    u = rng.integers(0, 256, 30).astype(np.uint8)
    p.write_bytes(_wav_bytes(1, 1, 8000, 8, u.tobytes()))
    x, rate = WAV.read_wav(p)
    assert rate == 8000 and np.array_equal(x[:, 0], ors.to_s16(u, "u8"))
    # not PCM (mu-law tag 7) and not RIFF -> the reference's own conversion hint
    p.write_bytes(_wav_bytes(7, 1, 8000, 8, u.tobytes()))
    with pytest.raises(WAV.AudioFormatError, match="ffmpeg -i IN -ar 16000 -ac 1 -f wav -acodec pcm_s16le"):
        WAV.read_wav(p)
    p.write_bytes(b"ID3\x00 not a wave file")
    with pytest.raises(WAV.AudioFormatError, match="not a RIFF/WAVE"):
        WAV.read_wav(p)
    # the strict reader still refuses anything outside the contract
    p.write_bytes(_wav_bytes(1, 2, 44100, 16, s.astype("<i2").tobytes()))
    with pytest.raises(WAV.AudioFormatError):
        WAV.read_wav_s16(p)
