"""End-to-end through the drop-in boundary on a real GPU: Backend.enroll_speaker / identify_speaker /
verify_speaker (the EmbeddingBackend contract, speaker_detection_backends/base.py:107-180) and the
speaker-assign compatible driver, checked against the CPU oracle running the same pipeline."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import sub
from oracle import ecapa as oecapa
from oracle import fbank as ofbank
from oracle import scoring as oscoring

pytestmark = pytest.mark.gpu

wav = sub("wav")
W = sub("weights")


def _voice(seed, seconds, f0):
    """Synthetic 'speaker': a harmonic stack with a speaker-specific pitch and formant tilt + noise."""
    rng = np.random.default_rng(seed)
    t = np.arange(int(16000 * seconds)) / 16000.0
    x = sum((0.5 / h ** (1.0 + 0.2 * (seed % 3))) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) for h in range(1, 12))
    x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 3.1 * t)) + rng.normal(0, 0.02, t.shape)
    return np.clip(np.round(x / np.abs(x).max() * 0.5 * 32767), -32768, 32767).astype(np.int16)


def _oracle_embed(pcm_windows, weights=None):
    feats = torch.from_numpy(ofbank.fbank(pcm_windows))
    return oecapa.l2_normalise(oecapa.EcapaOracle(weights if weights is not None else W.synthetic_weights(0), "bf16", torch.float64).embed(feats).numpy())


@pytest.mark.parametrize("bias_correction", ["0", "1"])
def test_enroll_identify_verify_roundtrip(tmp_path, monkeypatch, bias_correction):
    """Both settings of the bf16 weight-rounding bias correction through the plug-in class: "1" is what ships (calibration pass -> 29 patched bias
    slots -> cache entry "0c"), "0" the plain model.  The oracle is the bf16 layer-boundary model of the engine's EFFECTIVE weights either way."""
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path / "store"))
    monkeypatch.setenv("SDK_BIAS_CORRECTION", bias_correction)
    monkeypatch.setenv("SDK_CACHE_DIR", str(tmp_path / "cache"))
    be = sub("backend").Backend()
    assert be.engine().bias_correction is (bias_correction == "1")
    eff = be.engine().effective_weights()
    plain = W.synthetic_weights(0)
    assert any(not np.array_equal(eff[k], plain[k]) for k in plain) is (bias_correction == "1")
    assert any((tmp_path / "cache").glob("*.p0c.npy" if bias_correction == "1" else "*.p0.npy"))        # the entry this setting builds
    voices = {"alice": 140.0, "bob": 95.0, "carol": 210.0}
    profiles, oracle_vecs = [], {}
    for i, (sid, f0) in enumerate(voices.items()):
        path = tmp_path / f"enroll_{sid}.wav"
        wav.write_wav_s16(path, _voice(10 + i, 6.0, f0))
        rec = be.enroll_speaker(path, [(0.5, 5.5)])
        assert rec["external_id"].startswith("npy:") and rec["model_version"].startswith("mi355x-ecapa1024-")
        assert os.path.exists(rec["file"]) and rec["n_windows"] == 4
        profiles.append({"id": sid, "names": {"default": sid.title()}, "embeddings": {"mi355x": [
            {"id": f"emb-{sid}", "external_id": rec["external_id"], "model_version": rec["model_version"], "trust_level": "high"}]}})
        # oracle enrollment on the same windows
        pcm, _ = wav.cut_windows(wav.read_wav_s16(path), [(0.5, 5.5)])
        e = _oracle_embed(pcm, eff).astype(np.float64).mean(0)
        oracle_vecs[sid] = (e / np.linalg.norm(e)).astype(np.float32)
        stored = np.load(rec["file"])
        assert float(stored @ oracle_vecs[sid]) > 1 - 1e-4, "stored enrollment vector vs oracle"

    # test recording: bob for 4 s, then alice for 4 s (different noise seeds than enrollment)
    test = np.concatenate([_voice(40, 4.0, 95.0), _voice(41, 4.0, 140.0)])
    tpath = tmp_path / "meeting.wav"
    wav.write_wav_s16(tpath, test)
    rows = be.identify_speaker(tpath, profiles, threshold=0.354)
    assert {r["speaker_id"] for r in rows} >= {"alice", "bob"}
    assert all(set(r) >= {"speaker_id", "similarity", "confidence", "embedding_id", "segment"} for r in rows)

    # oracle: same windows, same profiles, same aggregation
    pcm, spans = wav.cut_windows(wav.read_wav_s16(tpath), None)
    Eo = _oracle_embed(pcm, eff)
    Pm = oecapa.l2_normalise(np.stack([np.load(str(tmp_path / "store/embeddings/by-hash" / (p["embeddings"]["mi355x"][0]["external_id"][4:] + ".npy")))
                                       for p in profiles]))
    oidx, osc = oscoring.affinity_topk(Eo, Pm, 1)
    # GPU per-window assignments (recomputed through the same public pieces)
    E, Eb, re = be.embed_windows(pcm)
    batch = sub("store").load_profile_batch(profiles, "mi355x", model_prefix="mi355x-")
    gidx, gsc = be.score_windows(E, Eb, re, batch)
    cos = (E.cpu().numpy().astype(np.float64) * Eo).sum(1)
    assert (cos > 1 - 1e-4).all(), cos
    margin = np.sort(oscoring.affinity(Eo, Pm), axis=1)
    clear = (margin[:, -1] - margin[:, -2]) > 1e-3          # windows whose oracle decision is not a near-tie
    assert np.array_equal(gidx[clear, 0], oidx[clear, 0]), (gidx[:, 0], oidx[:, 0])
    assert np.abs(gsc[:, 0] - osc[:, 0])[clear].max() < 5e-4   # end-to-end (embedding + score) tolerance; score-only is 1e-5

    ok = be.verify_speaker(tpath, profiles[0], threshold=0.354)
    assert ok["match"] is True and ok["confidence"] == ok["similarity"] and ok["embedding_id"] == "emb-alice"
    assert be.verify_speaker(tpath, profiles[2], threshold=0.99) == {"match": False, "similarity": 0.0, "confidence": 0.0, "embedding_id": None}

    # speaker-assign compatible driver on top (in-process rows instead of a subprocess per label)
    db = tmp_path / "store" / "db"
    db.mkdir(parents=True, exist_ok=True)
    for p in profiles:
        (db / f"{p['id']}.json").write_text(json.dumps(p))
    transcript = tmp_path / "t.json"
    transcript.write_text(json.dumps({"results": [
        {"type": "word", "start_time": 0.2, "end_time": 3.8, "alternatives": [{"content": "hello", "speaker": "S1"}]},
        {"type": "word", "start_time": 4.2, "end_time": 7.8, "alternatives": [{"content": "there", "speaker": "S2"}]}]}))
    asg, ident = sub("assign"), sub("identify")
    rows_fn = ident.make_rows_fn(tpath, per_label=True, backend=be)
    out = asg.assign_recording(tpath, transcript, rows_fn=rows_fn, use_embeddings=True, threshold=0.1)
    assert out["mappings"]["S1"]["speaker_id"] == "bob" and out["mappings"]["S2"]["speaker_id"] == "alice", out["mappings"]
    assert out["mappings"]["S1"]["signals"][0]["trust_level"] == "high" and out["mappings"]["S1"]["signals"][0]["backend"] == "mi355x"


def test_pipeline_run_shard_single_rank(engine):
    """configs #2 + #5 glued: embed -> assign vs profiles -> cluster the same segments; every stage vs the oracle."""
    from oracle import spectral as ospec
    P = sub("pipeline")
    # 3 synthetic 'speakers' x 8 two-second windows each
    pcm = np.stack([_voice(100 + s * 17 + i, 2.0, f0)[:32000] for s, f0 in enumerate((95.0, 150.0, 220.0)) for i in range(8)])
    rng = np.random.default_rng(5)
    profiles = oecapa.l2_normalise(rng.standard_normal((7, 192)).astype(np.float32))
    res = P.run_shard(engine, torch.from_numpy(pcm).cuda(), torch.from_numpy(profiles).cuda(), k=1, threshold=-1.0, n_clusters=3, cluster_iters=20)
    Eo = _oracle_embed(pcm)
    Eg = res.embeddings.cpu().numpy()
    assert ((Eg.astype(np.float64) * Eo).sum(1) > 1 - 1e-4).all()
    oidx, osc = oscoring.affinity_topk(Eg, profiles, 1)
    assert np.array_equal(res.best_profile, oidx) and np.abs(res.best_score - osc).max() <= 1e-5
    olab, _ = ospec.spectral_cluster(oecapa.to_bf16_f32(Eg), 3, n_iter=20, n_kmeans=20)
    assert np.array_equal(res.cluster_labels, olab)


def test_graph_replay_matches_eager(engine):
    """The captured-HIP-graph form of the embedding path returns exactly what the eager launches return."""
    for B in (1, 5):
        pcm = torch.from_numpy(np.stack([_voice(200 + i, 2.0, 120.0 + 10 * i)[:32000] for i in range(B)])).cuda()
        E0, Eb0, r0 = [t.clone() for t in engine.embed_pcm(pcm)]
        E1, Eb1, r1 = engine.embed_pcm_graph(pcm)
        torch.cuda.synchronize()
        assert torch.equal(E0, E1) and torch.equal(Eb0.float(), Eb1.float()) and torch.equal(r0, r1)
        pcm2 = torch.roll(pcm, 1234, dims=1)
        E2, _, _ = engine.embed_pcm_graph(pcm2)          # replay with new input
        torch.cuda.synchronize()
        assert torch.equal(E2, engine.embed_pcm(pcm2)[0])


def test_embed_windows_in_bounded_batches(monkeypatch):
    """Backend.embed_windows cuts a long recording into SDK_MAX_BATCH-window batches; the embeddings
    must not depend on the cut beyond the batch-invariance tolerance written in test_config2_full_batch_properties (|dE| <= 1e-3 on
    unit vectors, cos >= 1 - 1e-5: per-segment fp32 statistics are summed in another grouping when a segment starts on another tile row)."""
    import torch
    B = sub("backend")
    rng = np.random.default_rng(7)
    t = np.arange(32000) / 16000.0
    pcm = np.stack([np.clip(np.round(3000 * np.sin(2 * np.pi * (90 + 17 * i) * t) + 1200 * np.sin(2 * np.pi * (300 + 41 * i) * t)
                                     + rng.normal(0, 700, t.shape)), -32768, 32767).astype(np.int16) for i in range(10)])
    be = B.Backend()
    monkeypatch.delenv("SDK_MAX_BATCH", raising=False)
    E1, Eb1, r1 = be.embed_windows(pcm)
    monkeypatch.setenv("SDK_MAX_BATCH", "3")                 # 3 + 3 + 3 + 1
    E2, Eb2, r2 = be.embed_windows(pcm)
    torch.cuda.synchronize()
    assert E2.shape == (10, 192) and Eb2.shape == (10, 192) and r2.shape == (10,)
    assert torch.equal(E1[:3], E2[:3])                       # the first batch starts on the same tile rows: bit-identical
    dmax = float((E1 - E2).abs().max())
    cmin = float((E1.double() * E2.double()).sum(1).min())
    assert dmax < 1e-3 and cmin > 1 - 1e-5, (dmax, cmin)
    monkeypatch.setenv("SDK_MAX_BATCH", "1")
    E3 = be.embed_windows(pcm)[0]
    assert float((E1 - E3).abs().max()) < 1e-3


PACK_CHILD = r"""
import importlib, json, sys
sys.path.insert(0, sys.argv[1])
B = importlib.import_module(sys.argv[2] + ".backend")
be = B.Backend()
cands = json.load(open(sys.argv[3]))
rows = be.identify_speaker(sys.argv[4], cands, threshold=-1.0)
print(json.dumps({"rows": rows, "from_pack": bool(be.last_batch.from_pack), "torch": "torch" in sys.modules}))
"""


def test_profile_pack_hit_scores_bit_identically_to_the_miss_that_built_it(tmp_path, monkeypatch):
    """k7: the first identify over a candidate set loads every embedding file, normalises on the device and publishes the set's
    pack; a later Backend - and the torch-free host path in another process - maps that ONE file, uploads the normalised copies as packed and must
    return the same rows bit for bit."""
    import subprocess
    import sys
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path / "store"))
    monkeypatch.setenv("SDK_CACHE_DIR", str(tmp_path / "cache"))
    store = sub("store")
    be = sub("backend").Backend()
    path = tmp_path / "meeting.wav"
    wav.write_wav_s16(path, np.concatenate([_voice(40, 4.0, 95.0), _voice(41, 4.0, 140.0)]))
    enr = tmp_path / "enr.wav"
    wav.write_wav_s16(enr, _voice(10, 6.0, 140.0))
    rec = be.enroll_speaker(enr)
    mv = rec["model_version"]
    cands = [{"id": "alice", "embeddings": {"mi355x": [{"id": "emb-alice", "external_id": rec["external_id"], "model_version": mv, "trust_level": "high"}]}}]
    rng = np.random.default_rng(3)
    for i in range(70):
        v = rng.standard_normal(192).astype(np.float32) * np.float32(1.0 + 0.01 * i)         # not unit length: the device normalisation matters
        cands.append({"id": f"spk{i}", "embeddings": {"mi355x": [{"id": f"emb-{i}", "external_id": store.save_vector(v), "model_version": mv}]}})
    rows_miss = be.identify_speaker(path, cands, threshold=-1.0)
    assert be.last_batch.from_pack is False and be.last_batch.pack_ref is None                 # published
    assert len(list((tmp_path / "store" / "embeddings" / "packs").glob("pack-*.npy"))) == 1
    be2 = sub("backend").Backend()
    rows_hit = be2.identify_speaker(path, cands, threshold=-1.0)
    assert be2.last_batch.from_pack is True
    assert rows_hit == rows_miss and rows_miss[0]["speaker_id"] == "alice"
    monkeypatch.setenv("SDK_PROFILE_PACK", "0")                                                # and both equal the plain per-file path
    be3 = sub("backend").Backend()
    assert be3.identify_speaker(path, cands, threshold=-1.0) == rows_miss and be3.last_batch.from_pack is False
    monkeypatch.delenv("SDK_PROFILE_PACK")
    (tmp_path / "cands.json").write_text(json.dumps(cands))
    for lite in ("1", "0"):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("SDK_NO_TORCH", None)
        if lite == "1":
            env["SDK_NO_TORCH"] = "1"
        r = subprocess.run([sys.executable, "-c", PACK_CHILD, str(sub("backend").__file__.rsplit("/", 2)[0]), sub("backend").__package__,
                            str(tmp_path / "cands.json"), str(path)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert got["from_pack"] is True and got["torch"] is (lite == "0")
        assert json.loads(json.dumps(rows_miss)) == got["rows"]
