"""SURVEY.md §8f-1 on a real GPU: per-sentence identify over the committed Speechmatics fixture (10 `is_eos` sentences, 5 Alice /
5 Bob, the speaker change at 5.36 s) with two enrolled stand-in voices: ONE bucketed GPU pass per recording over true-length
windows, each label scored on its own windows, checked against the CPU oracle run on the same cuts."""
import json

import numpy as np
import pytest
import torch

from conftest import ROOT, sub
from oracle import ecapa as oecapa
from oracle import fbank as ofbank
from oracle import scoring as oscoring

pytestmark = pytest.mark.gpu

wav = sub("wav")
W = sub("weights")
FIXTURE = ROOT / "tests" / "golden" / "test_001-two-speakers.wav.speechmatics.json"


def _voice(seed, seconds, f0):
    rng = np.random.default_rng(seed)
    t = np.arange(int(round(16000 * seconds))) / 16000.0
    x = sum((0.5 / h ** (1.0 + 0.2 * (seed % 3))) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) for h in range(1, 12))
    x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 3.1 * t)) + rng.normal(0, 0.02, t.shape)
    return np.clip(np.round(x / np.abs(x).max() * 0.5 * 32767), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("bias_correction", ["0", "1"])
def test_fixture_sentences_one_pass_per_recording(tmp_path, monkeypatch, bias_correction):
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path / "store"))
    monkeypatch.setenv("SDK_BIAS_CORRECTION", bias_correction)             # "1" = the shipped default; the oracle below runs on the engine's effective weights
    monkeypatch.setenv("SDK_CACHE_DIR", str(tmp_path / "cache"))
    be = sub("backend").Backend()
    seg, ident, asg, store = sub("segments"), sub("identify"), sub("assign"), sub("store")
    data = json.loads(FIXTURE.read_text())
    sents = seg.sentence_segments(data)
    f0 = {"Alice": 140.0, "Bob": 95.0}
    # the recording: each sentence is filled with its speaker's voice, the gaps with faint noise
    rec = np.random.default_rng(0).normal(0, 20, int(16000 * 11.6)).astype(np.int16)
    for i, s in enumerate(sents):
        a, b = round(s["start"] * 16000), round(s["end"] * 16000)
        rec[a:b] = _voice(300 + i, (b - a) / 16000.0, f0[s["speaker"]])[:b - a]
    tpath = tmp_path / "two_speakers.wav"
    wav.write_wav_s16(tpath, rec)
    # enrollment from separate clips, with true-length segment windows
    db = tmp_path / "store" / "db"
    db.mkdir(parents=True)
    profiles = []
    for i, (label, sid) in enumerate((("Alice", "alice"), ("Bob", "bob"))):
        p = tmp_path / f"enroll_{sid}.wav"
        wav.write_wav_s16(p, _voice(20 + i, 6.0, f0[label]))
        enr = be.enroll_speaker(p, [(0.5, 5.5), (5.6, 5.95)])            # 4 two-second windows + nothing for the 0.35-s range
        assert enr["n_windows"] == 4
        prof = {"id": sid, "names": {"default": label}, "embeddings": {"mi355x": [
            {"id": f"emb-{sid}", "external_id": enr["external_id"], "model_version": enr["model_version"], "trust_level": "high"}]}}
        (db / f"{sid}.json").write_text(json.dumps(prof))
        profiles.append(prof)

    calls, uploads = [], []
    real = be.engine().fbank_windows
    monkeypatch.setattr(be.engine(), "fbank_windows", lambda ds, n, dw, B, S, **kw: (calls.append((B, S)), real(ds, n, dw, B, S, **kw))[1])
    ing = be.engine().ingest()
    real_submit = ing.submit
    monkeypatch.setattr(ing, "submit", lambda smp, st, W, stream: (uploads.append((len(smp), len(st))), real_submit(smp, st, W, stream))[1])
    rows_fn = ident.make_rows_fn(tpath, per_label=True, backend=be, transcript=data)
    assert sorted(calls) == [(4, 24000), (6, 8000), (7, 16000)]            # the whole recording: one launch sequence per bucket
    assert uploads == [(len(rec), 17)]                                      # ... and ONE upload: the recording as it is + 17 window starts
    out = asg.assign_recording(tpath, FIXTURE, rows_fn=rows_fn, use_embeddings=True, threshold=0.1)
    assert len(calls) == 3                                                   # labels are served from that one pass
    assert out["mappings"]["Alice"]["speaker_id"] == "alice" and out["mappings"]["Bob"]["speaker_id"] == "bob", out["mappings"]
    assert not any(a < 5.36 < b for _, a, b in rows_fn.windows) and len(rows_fn.windows) == 17
    assert {lab for lab, _, _ in rows_fn.windows} == {"Alice", "Bob"}

    # the same cuts through the CPU oracle: fbank -> ECAPA-TDNN (bf16 layer-boundary model) -> L2 -> cosine argmax
    samples = wav.read_wav_s16(tpath)
    pcm_by_len, wins, dropped = wav.cut_ranges(samples, [(s["start"], s["end"]) for s in sents])
    assert dropped == [6]
    E, Eb, re, gw, _ = be.embed_ranges(samples, [(s["start"], s["end"]) for s in sents])
    assert [(ri, a, b) for ri, _, _, a, b in wins] == gw
    assert be.engine().bias_correction is (bias_correction == "1")
    orc = oecapa.EcapaOracle(be.engine().effective_weights(), "bf16", torch.float64)
    Eo = np.zeros((len(wins), 192), np.float32)
    for S, pcm in pcm_by_len.items():
        e = oecapa.l2_normalise(orc.embed(torch.from_numpy(ofbank.fbank(pcm))).numpy())
        for w, (_, S2, row, _, _) in enumerate(wins):
            if S2 == S:
                Eo[w] = e[row]
    cos = (E.cpu().numpy().astype(np.float64) * Eo).sum(1)
    assert (cos > 1 - 1e-4).all(), cos
    batch = store.load_profile_batch(profiles, "mi355x", model_prefix="mi355x-", model_version=be.model_version)
    gidx, gsc = be.score_windows(E, Eb, re, batch)
    Pm = oecapa.l2_normalise(batch.matrix)
    oidx, osc = oscoring.affinity_topk(Eo, Pm, 1)
    full = oscoring.affinity(Eo, Pm)
    clear = np.abs(full[:, 0] - full[:, 1]) > 1e-3
    assert clear.sum() >= 15 and np.array_equal(gidx[clear, 0], oidx[clear, 0])
    assert np.abs(gsc[:, 0] - osc[:, 0])[clear].max() < 5e-4
    want = np.array([0 if sents[ri]["speaker"] == "Alice" else 1 for ri, _, _ in gw])
    assert (oidx[:, 0] == want).mean() >= 0.9                                 # the stand-in voices are separable per sentence
