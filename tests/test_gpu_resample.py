"""Audio conversion to the AudioProfile on the GPU: `sdk_resample_s16` through the C-ABI must equal the
integer oracle bit for bit; then the plug-in boundary accepts a 48 kHz stereo recording of a voice and lands on the
same speaker as the 16 kHz mono original."""
import struct

import numpy as np
import pytest
import torch

from conftest import sub
from oracle import resample as ors

pytestmark = pytest.mark.gpu

wav = sub("wav")


@pytest.mark.parametrize("rate,ch,n", [(48000, 1, 48000), (44100, 2, 50001), (8000, 1, 4000), (22050, 2, 22050), (11025, 1, 9000),
                                       (96000, 6, 30000), (32000, 3, 12345), (16000, 2, 5000), (44100, 1, 1), (48000, 2, 7),
                                       (8000, 1, 3), (44100, 2, 89)])
def test_resample_bit_exact(engine, rate, ch, n):
    rng = np.random.default_rng(rate + ch + n)
    x = rng.integers(-32768, 32768, (n, ch)).astype(np.int16)       # full-scale noise: exercises saturation too
    t, L, M, K = ors.design_taps(rate, 16000)
    want = ors.resample_s16(x, t, L, M)
    got = engine.resample_s16(torch.from_numpy(x).cuda(), rate, 16000)
    torch.cuda.synchronize()
    assert got.dtype == torch.int16 and got.shape == (len(want),)
    assert np.array_equal(got.cpu().numpy(), want)
    if ch == 1:                                                      # [n] and [n, 1] are the same input
        got1 = engine.resample_s16(torch.from_numpy(x[:, 0].copy()).cuda(), rate, 16000)
        assert np.array_equal(got1.cpu().numpy(), want)


def test_resample_full_size_properties(engine):
    """One hour of 48 kHz stereo (691 MB) -> 16 kHz mono: constants stay constants, the length is exact, a sampled
    stretch equals the oracle run on that stretch alone (the filter only sees K/2 neighbours)."""
    n = 48000 * 3600
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randint(-20000, 20000, (n, 2), dtype=torch.int16, device="cuda", generator=g)
    x[1_000_000:2_000_000] = 777
    y = engine.resample_s16(x, 48000, 16000)
    torch.cuda.synchronize()
    assert y.shape == (n // 3,)
    assert (y[1_000_000 // 3 + 40:2_000_000 // 3 - 40] == 777).all()
    t, L, M, K = ors.design_taps(48000, 16000)
    a = 90_000_000                                                    # multiple of M: output index a / 3
    seg = x[a - K:a + 3000 + K].cpu().numpy()
    want = ors.resample_s16(seg, t, L, M)[K // 3:K // 3 + 1000]
    assert np.array_equal(y[a // 3:a // 3 + 1000].cpu().numpy(), want)


def _voice48(seed, seconds, f0):
    rng = np.random.default_rng(seed)
    tt = np.arange(int(48000 * seconds)) / 48000.0
    x = sum((0.5 / h ** 1.2) * np.sin(2 * np.pi * f0 * h * tt + rng.uniform(0, 6.28)) for h in range(1, 12))
    x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 3.1 * tt)) + rng.normal(0, 0.01, tt.shape)
    return x / np.abs(x).max() * 0.5


def test_backend_accepts_other_wave_layouts(tmp_path, monkeypatch):
    """enroll from a 48 kHz stereo float WAVE, identify on the 16 kHz mono s16 rendering of another take."""
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path / "store"))
    be = sub("backend").Backend()
    profiles = []
    for i, (sid, f0) in enumerate({"alice": 140.0, "bob": 95.0}.items()):
        v = _voice48(70 + i, 5.0, f0).astype("<f4")
        st = np.stack([v, v], axis=1)
        fmt = struct.pack("<HHIIHH", 3, 2, 48000, 48000 * 8, 8, 32)
        body = b"WAVEfmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", st.nbytes) + st.tobytes()
        path = tmp_path / f"{sid}_48k.wav"
        path.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
        rec = be.enroll_speaker(path, [(0.5, 4.5)])
        # the decoded audio is what the oracle's converter yields for this file
        x, rate = wav.read_wav(path)
        t, L, M, K = ors.design_taps(rate, 16000)
        assert np.array_equal(wav.decode_to_profile(path, be.engine()), ors.resample_s16(x, t, L, M))
        profiles.append({"id": sid, "embeddings": {"mi355x": [{"id": f"emb-{sid}", "external_id": rec["external_id"],
                                                              "model_version": rec["model_version"], "trust_level": "high"}]}})
    v16 = _voice48(90, 4.0, 95.0)[::3]                                # bob again, plain decimation is fine for a test tone stack < 1.1 kHz
    tpath = tmp_path / "take.wav"
    wav.write_wav_s16(tpath, np.round(v16 * 32767).astype(np.int16))
    rows = be.identify_speaker(tpath, profiles, threshold=0.0)
    assert rows and rows[0]["speaker_id"] == "bob", rows
