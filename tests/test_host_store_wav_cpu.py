"""Host logic either side of the GPU path: WAV contract reader, window cutting, embeddings store +
batch loader (k7), result aggregation, registry."""
import json
import os
from pathlib import Path

import numpy as np
import pytest

from conftest import PKG, ROOT, sub

wav = sub("wav")
store = sub("store")
backend = sub("backend")
api = sub("plugin_api")


def test_wav_roundtrip_and_contract(tmp_path):
    x = (np.random.default_rng(0).standard_normal(16000) * 3000).astype(np.int16)
    p = tmp_path / "a.wav"
    wav.write_wav_s16(p, x)
    assert np.array_equal(wav.read_wav_s16(p), x)
    wav.write_wav_s16(tmp_path / "b.wav", x, rate=8000)
    with pytest.raises(wav.AudioFormatError, match="-ar 16000 -ac 1 -f wav -acodec pcm_s16le"):
        wav.read_wav_s16(tmp_path / "b.wav")
    (tmp_path / "c.wav").write_bytes(b"not a wav at all")
    with pytest.raises(wav.AudioFormatError):
        wav.read_wav_s16(tmp_path / "c.wav")


def test_cut_windows():
    x = np.arange(16000 * 10, dtype=np.int16)
    pcm, spans = wav.cut_windows(x, None)                       # whole file: 2-s windows, 1-s hop
    assert pcm.shape == (9, 32000) and spans[0] == (0.0, 2.0) and spans[-1] == (8.0, 10.0)
    assert np.array_equal(pcm[3], x[48000:80000])
    pcm, spans = wav.cut_windows(x, [(1.0, 1.2), (3.0, 4.0), (5.0, 9.5)])
    assert spans[0] == (2.5, 4.5)                              # 1-s segment widened symmetrically to the 2-s window
    assert len(spans) == 1 + 4 and spans[-1] == (7.5, 9.5)    # (1.0,1.2) dropped (< 0.5 s); tail window flush with the end
    short = np.ones(8000, dtype=np.int16)
    pcm, spans = wav.cut_windows(short, None)
    assert pcm.shape == (1, 32000) and pcm[0, 8000:].sum() == 0 and spans == [(0.0, 0.5)]


def test_cut_ranges_never_leaves_a_sentence():
    """the committed Speechmatics fixture's 10 `is_eos` sentences (5 Alice, 5 Bob).  Every window lies
    inside its own sentence - so none crosses the speaker change at 5.36 s, which the old 2-s widening did
    (Alice's (4.64, 5.36) was embedded from 4.0-6.0 s) - lengths come from the bucket table, nothing is padded."""
    seg = sub("segments")
    data = json.loads((ROOT / "tests" / "golden" / "test_001-two-speakers.wav.speechmatics.json").read_text())
    sents = seg.sentence_segments(data)
    assert [s["speaker"] for s in sents] == ["Alice"] * 5 + ["Bob"] * 5
    x = np.arange(16000 * 12, dtype=np.int32).astype(np.int16)
    pcm_by_len, wins, dropped = wav.cut_ranges(x, [(s["start"], s["end"]) for s in sents])
    assert dropped == [6]                                            # Bob's 0.48-s sentence is below the smallest bucket
    assert set(pcm_by_len) <= {8000, 16000, 24000, 32000} and sum(len(v) for v in pcm_by_len.values()) == len(wins)
    covered = {}
    for ri, S, row, a, b in wins:
        s0, s1 = sents[ri]["start"], sents[ri]["end"]
        assert s0 - 1e-9 <= a and b <= s1 + 1e-9 and abs((b - a) * 16000 - S) < 1e-6          # inside the sentence, true length
        assert S <= round((s1 - s0) * 16000) and (S == 32000 or round((s1 - s0) * 16000) < S + 8000)   # the largest bucket that fits
        assert np.array_equal(pcm_by_len[S][row], x[round(a * 16000):round(a * 16000) + S])
        covered.setdefault(ri, []).append((a, b))
    for ri, spans in covered.items():                                 # a sentence is covered end to end
        assert abs(min(a for a, _ in spans) - sents[ri]["start"]) < 1e-9 and abs(max(b for _, b in spans) - sents[ri]["end"]) < 1e-9
    assert not any(a < 5.36 < b for _, _, _, a, b in wins)
    # long ranges: 2-s windows at the hop, the last flush with the end; and the old cut_windows slip (a range's tail test
    # looked at the PREVIOUS range's last start when the current range produced none) stays fixed
    _, w2, _ = wav.cut_ranges(x, [(1.0, 6.3)])
    assert [(round(a, 2), round(b, 2)) for _, _, _, a, b in w2] == [(1.0, 3.0), (2.0, 4.0), (3.0, 5.0), (4.0, 6.0), (4.3, 6.3)]
    pcm, spans = wav.cut_windows(x, [(0.0, 4.5), (6.0, 8.0)])
    assert spans == [(0.0, 2.0), (1.0, 3.0), (2.0, 4.0), (2.5, 4.5), (6.0, 8.0)]


def test_cut_ranges_properties_on_random_ranges():
    """Property form of the rule above on arbitrary (also overlapping, out-of-file, tiny) ranges: every window is a true-length bucket,
    lies inside its range clipped to the file, the windows of a range cover it end to end without gaps, rows of the per-length
    batches are the file's samples, and exactly the ranges shorter than the smallest bucket are dropped."""
    from hypothesis import given, settings, strategies as st
    rate, n = 16000, 16000 * 20
    x = (np.arange(n, dtype=np.int64) * 7919 % 65536 - 32768).astype(np.int16)
    rng = st.tuples(st.floats(-1.0, 21.0, allow_nan=False), st.floats(0.0, 9.0, allow_nan=False)).map(lambda t: (t[0], t[0] + t[1]))

    @settings(max_examples=150, deadline=None)
    @given(st.lists(rng, min_size=0, max_size=12))
    def check(ranges):
        pcm_by_len, wins, dropped = wav.cut_ranges(x, ranges)
        assert set(pcm_by_len) <= {8000, 16000, 24000, 32000}
        assert sum(len(v) for v in pcm_by_len.values()) == len(wins)
        by_range = {}
        for ri, S, row, a, b in wins:
            lo, hi = max(0, round(ranges[ri][0] * rate)), min(n, round(ranges[ri][1] * rate))
            ia = round(a * rate)
            assert lo <= ia and ia + S <= hi and abs((b - a) * rate - S) < 1e-6
            assert S == max(q for q in (8000, 16000, 24000, 32000) if q <= hi - lo)
            assert np.array_equal(pcm_by_len[S][row], x[ia:ia + S])
            by_range.setdefault(ri, []).append((ia, ia + S))
        for ri, (s0, e0) in enumerate(ranges):
            lo, hi = max(0, round(s0 * rate)), min(n, round(e0 * rate))
            if hi - lo < 8000:
                assert ri in dropped and ri not in by_range
            else:
                spans = sorted(by_range[ri])
                assert ri not in dropped and spans[0][0] == lo and spans[-1][1] == hi
                assert all(spans[k + 1][0] <= spans[k][1] for k in range(len(spans) - 1))     # no gap between consecutive windows
    check()


def _profile(sid, recs):
    return {"id": sid, "names": {"default": sid.title()}, "embeddings": {"mi355x": recs}}


def test_store_and_batch_loader(tmp_path, monkeypatch):
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    rng = np.random.default_rng(1)
    vecs = rng.standard_normal((3, 192)).astype(np.float32)
    ext = [store.save_vector(v) for v in vecs]
    assert all(e.startswith("npy:") for e in ext) and store.save_vector(vecs[0]) == ext[0]     # content-addressed, idempotent
    assert np.array_equal(store.load_vector(ext[1]), vecs[1])
    cands = [
        _profile("alice", [{"id": "emb-a1", "external_id": ext[0], "model_version": "mi355x-ecapa1024-x", "trust_level": "high"},
                           {"id": "emb-a2", "external_id": ext[1], "model_version": "mi355x-ecapa1024-x", "trust_level": "low"}]),
        _profile("bob", [{"id": "emb-b1", "external_id": ext[2], "model_version": "mi355x-ecapa1024-x"},
                         {"id": "emb-b2", "external_id": "AD1NQVAB", "model_version": "speechmatics-v2"},
                         {"id": "emb-b3", "external_id": "npy:" + "0" * 24, "model_version": "mi355x-ecapa1024-x"}]),
        {"id": "carol", "embeddings": {}},
    ]
    batch = store.load_profile_batch(cands, "mi355x", model_prefix="mi355x-")
    assert batch.matrix.shape == (3, 192) and batch.speaker_ids == ["alice", "alice", "bob"]
    assert batch.embedding_ids == ["emb-a1", "emb-a2", "emb-b1"] and batch.trust_levels == ["high", "low", "unknown"]
    assert len(batch.skipped) == 2 and "speechmatics-v2" in batch.skipped[0]
    assert np.array_equal(batch.matrix, vecs)
    # per-speaker links: what speaker-report counts (speaker-report:292-294)
    assert sorted(p.name for p in (tmp_path / "embeddings" / "alice").glob("*.npy")) == ["emb-a1.npy", "emb-a2.npy"]
    assert batch.pack_ref is None and not batch.from_pack                                   # 3 rows: below SDK_PROFILE_PACK_MIN, no pack is offered
    with pytest.raises(ValueError):
        store.save_vector(np.zeros(10, np.float32))
    with pytest.raises(ValueError):
        store.vector_path("AD1NQVAB")


def test_batch_loader_skips_other_weights_and_hostile_keys(tmp_path, monkeypatch):
    """A local model's version carries its weights digest: a vector enrolled under weights A must not be scored against
    embeddings made with weights B; and an `external_id` from a user-editable db/*.json never leaves by-hash/."""
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    vecs = np.random.default_rng(2).standard_normal((2, 192)).astype(np.float32)
    ext = [store.save_vector(v) for v in vecs]
    outside = tmp_path / "embeddings" / "secret.npy"
    np.save(outside, vecs[0])
    cands = [_profile("alice", [{"id": "emb-a1", "external_id": ext[0], "model_version": "mi355x-ecapa1024-AAAA"},
                                {"id": "emb-a2", "external_id": ext[1], "model_version": "mi355x-ecapa1024-BBBB"},
                                {"id": "emb-a3", "external_id": "npy:../secret", "model_version": "mi355x-ecapa1024-BBBB"},
                                {"id": "emb-a4", "external_id": "npy:" + "g" * 24, "model_version": "mi355x-ecapa1024-BBBB"}])]
    loose = store.load_profile_batch(cands, "mi355x", model_prefix="mi355x-", link=False)
    assert loose.embedding_ids == ["emb-a1", "emb-a2"]                       # the toolkit's prefix rule alone lets both through
    batch = store.load_profile_batch(cands, "mi355x", model_prefix="mi355x-", model_version="mi355x-ecapa1024-BBBB")
    assert batch.embedding_ids == ["emb-a2"] and np.array_equal(batch.matrix[0], vecs[1])
    why = " | ".join(batch.skipped)
    assert "enrolled under mi355x-ecapa1024-AAAA" in why and "re-enroll" in why and why.count("malformed mi355x external_id") == 2
    assert not (tmp_path / "embeddings" / "alice" / "emb-a3.npy").exists()   # nothing was linked out of the hostile key
    for bad in ("npy:../../x", "npy:", "npy:ABCDEF0123456789ABCDEF01", "npy:0123456789abcdef0123456/"):
        with pytest.raises(ValueError):
            store.vector_path(bad)


def test_backend_identify_uses_the_exact_model_version(tmp_path, monkeypatch):
    """Enroll under weights A, identify under weights B: every stored vector is refused, with the reason on stderr, before
    any GPU work (so this runs without a device)."""
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    be = backend.Backend()
    be._digest = "bbbbbbbbbbbb"                                               # the loaded weights
    ext = store.save_vector(np.random.default_rng(3).standard_normal(192).astype(np.float32))
    cands = [_profile("alice", [{"id": "emb-a1", "external_id": ext, "model_version": "mi355x-ecapa1024-aaaaaaaaaaaa"}])]
    # nothing usable although candidates were offered: loud - the CLI turns the exception into "Error during
    # identification: ..." + rc 1 (speaker_detection:1072-1074), not into an empty "no match"
    with pytest.raises(ValueError, match="1 of 1 enrolled embeddings are unusable .*re-enroll"):
        be.identify_speaker(tmp_path / "missing.wav", cands)
    assert be.identify_speaker(tmp_path / "missing.wav", [{"id": "carol", "embeddings": {}}]) == []   # no candidates at all stays quiet
    assert be.check_embedding_compatibility(cands[0]["embeddings"]["mi355x"][0])["compatible"]   # the toolkit's own prefix rule still says yes


def test_make_rows_fn_is_loud_when_every_vector_is_stale(tmp_path, monkeypatch, capsys):
    """The in-process identify (identify.make_rows_fn) with a database enrolled under other weights: an empty rows_fn AND one
    summary line on stderr, in both the whole-recording and the per-label mode - `assign` must not silently lose its signal."""
    import json
    identify = sub("identify")
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    (tmp_path / "db").mkdir()
    ext = store.save_vector(np.random.default_rng(4).standard_normal(192).astype(np.float32))
    prof = _profile("alice", [{"id": "emb-a1", "external_id": ext, "model_version": "mi355x-ecapa1024-aaaaaaaaaaaa", "trust_level": "high"}])
    (tmp_path / "db" / "alice.json").write_text(json.dumps(prof))
    wav_path = tmp_path / "a.wav"
    wav_path.write_bytes(b"RIFF")                                            # only its existence is checked before the batch is loaded
    be = backend.Backend()
    be._digest = "bbbbbbbbbbbb"
    for per_label in (False, True):
        fn = identify.make_rows_fn(wav_path, per_label=per_label, backend=be)
        assert fn("S1", [{"start": 0.0, "end": 2.0}]) == []
        err = capsys.readouterr().err
        assert "Error during identification: 1 of 1 enrolled embeddings are unusable" in err and "re-enroll" in err
        assert "enrolled under mi355x-ecapa1024-aaaaaaaaaaaa" in err


def test_aggregate_matches():
    batch = store.ProfileBatch(np.zeros((3, 192), np.float32), ["alice", "alice", "bob"], ["emb-a1", "emb-a2", "emb-b1"], ["high", "low", "high"])
    spans = [(0, 2), (1, 3), (2, 4), (3, 5), (4, 6)]
    idx = np.array([0, 1, 1, 2, 2])
    sc = np.array([0.9, 0.5, 0.7, 0.2, 0.8], np.float32)
    rows = backend.aggregate_matches(idx, sc, spans, batch, 0.354)
    assert [r["speaker_id"] for r in rows] == ["bob", "alice"]                 # bob: mean(0.8) ; alice: mean(0.9,0.5,0.7)=0.7
    assert rows[1]["embedding_id"] == "emb-a2" and rows[1]["n_segments"] == 3 and rows[1]["segment"] == (0, 4)
    assert abs(rows[1]["similarity"] - np.mean(np.array([0.9, 0.5, 0.7], np.float32).astype(np.float64))) < 1e-12
    assert rows[0]["confidence"] == rows[0]["similarity"]
    assert backend.aggregate_matches(idx, sc, spans, batch, 0.95) == []


def test_registry(monkeypatch, tmp_path):
    api.reload_backends_config()
    monkeypatch.delenv("SPEAKER_BACKENDS_CONFIG", raising=False)
    assert api.list_backends() == ["mi355x"]
    be = api.get_backend("mi355x")
    assert be.name == "mi355x" and be.model_version.startswith("mi355x-ecapa1024-") and be.get_audio_profile().sample_rate == 16000
    assert be.check_embedding_compatibility({"model_version": be.model_version})["compatible"] is True
    with pytest.raises(ValueError, match="Unknown backend: nope. Available: mi355x"):
        api.get_backend("nope")
    cfg = tmp_path / "b.yaml"
    cfg.write_text("backends:\n  short: " + PKG + ".backend\n  long:\n    module: " + PKG + ".backend\n")
    monkeypatch.setenv("SPEAKER_BACKENDS_CONFIG", str(cfg))
    api.reload_backends_config()
    assert api.list_backends() == ["short", "long"] and api.get_backend("long").name == "mi355x"
    api.reload_backends_config()


def test_backend_plugs_into_the_reference_registry(monkeypatch):
    """Drop-in check against the real toolkit when it is mounted (build container only)."""
    ref = Path("/root/reference")
    if not ref.exists():
        pytest.skip("reference not mounted on this machine")
    import subprocess, sys
    code = (
        "import sys; sys.path[:0]=[%r,%r]\n"
        "from speaker_detection_backends import get_backend\n"
        "from speaker_detection_backends.base import EmbeddingBackend\n"
        "b=get_backend('mi355x'); assert isinstance(b, EmbeddingBackend), type(b).__mro__\n"
        "print(b.name, b.embedding_dim, b.get_audio_profile().sample_rate, b.model_version)\n" % (str(ref), str(ROOT)))
    env = dict(os.environ, SPEAKER_BACKENDS_CONFIG=str(ROOT / PKG / "backends.yaml"), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("mi355x 192 16000 mi355x-ecapa1024-")


def test_identify_rows_early_exits_mirror_cmd_identify(tmp_path, monkeypatch, capsys):
    """speaker_detection:1033-1057 / test_cli.py:594-634: missing audio, empty database, no embeddings for the backend -
    the in-process row provider reports the toolkit's messages and yields no signals (no GPU is touched)."""
    import json
    ident = sub("identify")
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    audio = tmp_path / "a.wav"
    wav.write_wav_s16(audio, np.zeros(16000, dtype=np.int16))
    assert ident.make_rows_fn(tmp_path / "nonexistent.wav")("S1", []) == []
    assert "Audio file not found" in capsys.readouterr().err
    assert ident.make_rows_fn(audio)("S1", []) == []
    assert "No speakers to match against." in capsys.readouterr().err
    (tmp_path / "db").mkdir()
    (tmp_path / "db" / "alice.json").write_text(json.dumps({"id": "alice", "tags": ["team"], "embeddings": {"speechmatics": [{"id": "e1", "external_id": "spk"}]}}))
    assert ident.make_rows_fn(audio)("S1", []) == []
    assert "No speakers with mi355x embeddings." in capsys.readouterr().err
    assert ident.make_rows_fn(audio, tags=["other"])("S1", []) == []
    assert "No speakers to match against." in capsys.readouterr().err


def test_packed_blob_cache_roundtrip_and_backend_digest(tmp_path, monkeypatch):
    """weights_cache.py (cold start): the packed blob + digest of a weight set are found by a key that needs no pass
    over the weights; a second process takes model_version from the entry without generating / hashing 20.8 M parameters; a damaged,
    foreign-format or disabled cache is ignored, never trusted."""
    import json
    wc, W, WP = sub("weights_cache"), sub("weights"), sub("weights_pack")
    monkeypatch.setenv("SDK_CACHE_DIR", str(tmp_path / "cache"))
    monkeypatch.delenv("SDK_ECAPA_WEIGHTS", raising=False)
    cfg = W.EcapaConfig(channels=256, mfa_channels=768)
    w = W.synthetic_weights(9, cfg)
    key = wc.key_for_seed(9, cfg)
    assert key != wc.key_for_seed(8, cfg) and wc.load_blob(key, 0) is None and wc.load_meta(key) is None
    for prec in (0, 1):
        blob, f = WP.pack_weights(w, cfg, precision=prec)
        wc.store(key, "abc123abc123", prec, blob, f)
        got, gf = wc.load_blob(key, prec)
        assert np.array_equal(np.asarray(got), blob) and gf == f and isinstance(got, np.memmap)
    assert wc.load_meta(key)["digest"] == "abc123abc123" and set(wc.load_meta(key)["fields"]) == {"0", "1"}
    blob0, f0 = WP.pack_weights(w, cfg, precision=0)
    wc.store(key, "abc123abc123", "0c", blob0, f0)                          # the bias-corrected blob of the default mode: its own entry
    got, gf = wc.load_blob(key, "0c")
    assert np.array_equal(np.asarray(got), blob0) and gf["precision"] == 0 and set(wc.load_meta(key)["fields"]) == {"0", "1", "0c"}
    # a flipped byte in the blob, or an offset table that points outside it, is a miss (the offsets are dereferenced on the device)
    fn = tmp_path / "cache" / f"{key}.p1.npy"
    raw = bytearray(fn.read_bytes()); raw[-7] ^= 0x40; fn.write_bytes(bytes(raw))
    assert wc.load_blob(key, 1) is None and wc.load_blob(key, 0) is not None
    raw[-7] ^= 0x40; fn.write_bytes(bytes(raw))
    assert wc.load_blob(key, 1) is not None
    meta = json.loads((tmp_path / "cache" / f"{key}.json").read_text())
    good = list(meta["fields"]["1"]["off"])
    for bad_off in (meta["fields"]["1"]["_bytes"], 128, -2):
        meta["fields"]["1"]["off"][5] = bad_off
        (tmp_path / "cache" / f"{key}.json").write_text(json.dumps(meta))
        assert wc.load_blob(key, 1) is None
    meta["fields"]["1"]["off"] = good
    (tmp_path / "cache" / f"{key}.json").write_text(json.dumps(meta))
    assert wc.load_blob(key, 1) is not None
    # a truncated blob or another format version is a miss
    meta = json.loads((tmp_path / "cache" / f"{key}.json").read_text())
    meta["fields"]["0"]["_bytes"] += 1
    (tmp_path / "cache" / f"{key}.json").write_text(json.dumps(meta))
    assert wc.load_blob(key, 0) is None and wc.load_blob(key, 1) is not None
    meta["format"] = -1
    (tmp_path / "cache" / f"{key}.json").write_text(json.dumps(meta))
    assert wc.load_meta(key) is None and wc.load_blob(key, 1) is None
    # file-keyed entries change with the file
    path = tmp_path / "w.npz"
    W.save_weights(path, w)
    k1 = wc.key_for_file(str(path))
    os.utime(path, ns=(1, 1))
    assert wc.key_for_file(str(path)) != k1
    # the backend's model_version comes from the cache entry when there is one (no weights are generated) ...
    be_key = wc.key_for_seed(0, W.DEFAULT_CONFIG)
    (tmp_path / "cache" / f"{be_key}.json").write_text(json.dumps({"format": wc.FORMAT, "digest": "feedfeedfeed", "fields": {}}))
    be = backend.Backend()
    assert be.model_version == "mi355x-ecapa1024-feedfeedfeed" and be._weights is None and be._cache_hit
    # ... and from the weights themselves when the cache is switched off
    monkeypatch.setenv("SDK_WEIGHTS_CACHE", "0")
    assert wc.load_meta(be_key) is None


# ---------------------------------------------------------------------------------------------------------------- packed profile matrix (k7)
def _fake_norm(mat):
    """Stand-in for sdk_l2norm on a CPU box: unit rows, their bf16 bits (round to nearest even) and the rounding residual norms."""
    E = (mat / np.linalg.norm(mat, axis=1, keepdims=True)).astype(np.float32)
    u = E.view(np.uint32)
    bits = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    back = (bits.astype(np.uint32) << 16).view(np.float32)
    return E, bits, np.linalg.norm(E - back, axis=1).astype(np.float32)


def _enrol(n, seed=0, mv="mi355x-ecapa1024-x"):
    rng = np.random.default_rng(seed)
    vecs = rng.standard_normal((n, 192)).astype(np.float32)
    vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
    cands = [_profile(f"spk{i:05d}", [{"id": f"emb-{i}", "external_id": store.save_vector(v), "model_version": mv, "trust_level": "high" if i % 2 else "low"}])
             for i, v in enumerate(vecs)]
    return vecs, cands


def test_profile_pack_miss_publish_hit(tmp_path, monkeypatch):
    """the first identify over a candidate set loads file by file and publishes the set's pack; the next process maps
    ONE file - same matrix, same side tables, the normalised copies exactly as published - and touches no per-embedding file."""
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    vecs, cands = _enrol(40)
    cands[3]["embeddings"]["mi355x"].append({"id": "emb-foreign", "external_id": "AD1NQVAB", "model_version": "speechmatics-v2"})
    kw = dict(model_prefix="mi355x-", model_version="mi355x-ecapa1024-x")
    miss = store.load_profile_batch(cands, "mi355x", **kw)
    assert not miss.from_pack and miss.norm is None and miss.pack_ref is not None and len(miss.skipped) == 1
    E, bits, r = _fake_norm(miss.matrix)
    path = store.publish_pack(miss, E, bits, r)
    assert path is not None and path.exists() and miss.pack_ref is None
    # the hit must not need the per-embedding files at all
    for f in (tmp_path / "embeddings" / "by-hash").glob("*.npy"):
        f.unlink()
    hit = store.load_profile_batch(cands, "mi355x", **kw)
    assert hit.from_pack and hit.pack_ref is None
    assert np.array_equal(np.asarray(hit.matrix), vecs) and hit.speaker_ids == miss.speaker_ids and hit.embedding_ids == miss.embedding_ids
    assert hit.trust_levels == miss.trust_levels and hit.skipped == miss.skipped
    assert all(np.array_equal(np.asarray(a), b) for a, b in zip(hit.norm, (E, bits, r)))
    assert hit.uid != miss.uid
    # SDK_PROFILE_PACK=0: the per-file loader, which now finds nothing
    monkeypatch.setenv("SDK_PROFILE_PACK", "0")
    off = store.load_profile_batch(cands, "mi355x", **kw)
    assert len(off) == 0 and not off.from_pack and off.pack_ref is None and len(off.skipped) == 41


def test_profile_pack_hit_makes_the_links_once_and_checks_its_blob(tmp_path, monkeypatch):
    """a pack built by a caller that did not link (link=False) must not leave the per-speaker links unmade for ever - the first hit
    with link=True runs the cheap pass once and records it in the side table; and a pack whose blob no longer matches its crc32 (a torn or
    overwritten file of the right size) is not served: the loader falls back to the per-embedding files."""
    import json
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    vecs, cands = _enrol(24)
    kw = dict(model_prefix="mi355x-", model_version="mi355x-ecapa1024-x")
    miss = store.load_profile_batch(cands, "mi355x", link=False, **kw)
    assert not miss.linked and not (tmp_path / "embeddings" / "spk00003").exists()
    npy = store.publish_pack(miss, *_fake_norm(miss.matrix))
    side = npy.with_suffix(".json")
    assert json.loads(side.read_text())["linked"] is False and isinstance(json.loads(side.read_text())["crc32"], int)
    hit0 = store.load_profile_batch(cands, "mi355x", link=False, **kw)
    assert hit0.from_pack and not (tmp_path / "embeddings" / "spk00003").exists()                      # a reader that does not link leaves it so
    hit1 = store.load_profile_batch(cands, "mi355x", link=True, **kw)
    assert hit1.from_pack and (tmp_path / "embeddings" / "spk00003" / "emb-3.npy").exists() and json.loads(side.read_text())["linked"] is True
    assert np.array_equal(np.load(tmp_path / "embeddings" / "spk00003" / "emb-3.npy"), vecs[3])
    calls = []
    monkeypatch.setattr(store, "adopt", lambda *a, **k: calls.append(a))
    assert store.load_profile_batch(cands, "mi355x", link=True, **kw).from_pack and calls == []       # once per pack
    # same size, other bytes: refused by the checksum
    blob = np.load(npy)
    blob[100] ^= 0xFF
    np.save(npy, blob)
    fallback = store.load_profile_batch(cands, "mi355x", link=False, **kw)
    assert not fallback.from_pack and len(fallback) == 24 and np.array_equal(fallback.matrix, vecs)


def test_profile_pack_goes_stale_with_the_candidate_set(tmp_path, monkeypatch):
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    vecs, cands = _enrol(20)
    kw = dict(model_prefix="mi355x-", model_version="mi355x-ecapa1024-x")
    b = store.load_profile_batch(cands, "mi355x", **kw)
    store.publish_pack(b, *_fake_norm(b.matrix))
    assert store.load_profile_batch(cands, "mi355x", **kw).from_pack
    import copy
    # (a) a trust level edited in db/*.json, (b) one speaker fewer, (c) a re-enrolment (new content hash), (d) other weights, (e) another order
    c1 = copy.deepcopy(cands); c1[5]["embeddings"]["mi355x"][0]["trust_level"] = "medium"
    c2 = cands[:-1]
    c3 = copy.deepcopy(cands); c3[0]["embeddings"]["mi355x"][0]["external_id"] = store.save_vector(-vecs[0])
    c5 = list(reversed(cands))
    for variant in (c1, c2, c3, c5):
        got = store.load_profile_batch(variant, "mi355x", **kw)
        assert not got.from_pack and got.pack_ref is not None
    assert not store.load_profile_batch(cands, "mi355x", model_prefix="mi355x-", model_version="mi355x-ecapa1024-y").from_pack
    assert np.array_equal(store.load_profile_batch(c3, "mi355x", **kw).matrix[0], -vecs[0])
    assert store.load_profile_batch(cands, "mi355x", **kw).from_pack                        # the original set's pack is still there
    # a truncated / foreign pack file is ignored, never trusted
    npy = next((tmp_path / "embeddings" / "packs").glob("pack-*.npy"))
    raw = npy.read_bytes()
    npy.write_bytes(raw[:len(raw) // 2])
    assert not store.load_profile_batch(cands, "mi355x", **kw).from_pack
    np.save(npy, np.zeros(7, np.float32))
    assert not store.load_profile_batch(cands, "mi355x", **kw).from_pack


def test_profile_pack_is_not_built_over_an_unreadable_vector(tmp_path, monkeypatch):
    """The digest is over the records' keys: a set with a missing vector file must not be packed, or the file turning up later (a store copied
    after its database) would stay "skipped" for ever."""
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    vecs, cands = _enrol(20)
    kw = dict(model_prefix="mi355x-", model_version="mi355x-ecapa1024-x")
    victim = next(f for f in (tmp_path / "embeddings" / "by-hash").glob("*.npy")
                  if np.array_equal(np.load(f, allow_pickle=False).astype(np.float32).reshape(-1), vecs[7]))
    raw = victim.read_bytes()
    victim.unlink()
    b = store.load_profile_batch(cands, "mi355x", **kw)
    assert len(b) == 19 and len(b.skipped) == 1 and b.pack_ref is None and not b.from_pack
    assert store.publish_pack(b, *_fake_norm(b.matrix)) is None
    victim.write_bytes(raw)
    b = store.load_profile_batch(cands, "mi355x", **kw)
    assert len(b) == 20 and not b.skipped and b.pack_ref is not None
    store.publish_pack(b, *_fake_norm(b.matrix))
    assert store.load_profile_batch(cands, "mi355x", **kw).from_pack


def test_profile_pack_is_bounded(tmp_path, monkeypatch):
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    _, cands = _enrol(60)
    kw = dict(model_prefix="mi355x-", model_version="mi355x-ecapa1024-x")
    for n in range(20, 20 + store.PACK_KEEP + 4):
        b = store.load_profile_batch(cands[:n], "mi355x", **kw)
        store.publish_pack(b, *_fake_norm(b.matrix))
        os.utime(next(iter(sorted((tmp_path / "embeddings" / "packs").glob("pack-*.json"), key=lambda q: q.stat().st_mtime, reverse=True))), (n, n))
    assert len(list((tmp_path / "embeddings" / "packs").glob("pack-*.json"))) == store.PACK_KEEP
    assert len(list((tmp_path / "embeddings" / "packs").glob("pack-*.npy"))) == store.PACK_KEEP


def _pack_builder(root, n, barrier, q):
    os.environ["SPEAKERS_EMBEDDINGS_DIR"] = root
    _, cands = _enrol(n)                                    # content-addressed: every process writes / finds the same by-hash files
    barrier.wait()
    b = store.load_profile_batch(cands, "mi355x", model_prefix="mi355x-", model_version="mi355x-ecapa1024-x")
    if not b.from_pack:
        store.publish_pack(b, *_fake_norm(b.matrix))
    again = store.load_profile_batch(cands, "mi355x", model_prefix="mi355x-", model_version="mi355x-ecapa1024-x")
    q.put((b.from_pack, again.from_pack, bool(np.array_equal(np.asarray(again.matrix), b.matrix)), len(again)))


def test_profile_pack_four_concurrent_builders(tmp_path):
    """speaker-process runs up to four CLI processes at once (speaker-process:627-629): four builders of the same set race; every one of them ends
    with a valid pack and the survivor is complete (atomic publish: matrix first, side table last)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(4), ctx.Queue()
    procs = [ctx.Process(target=_pack_builder, args=(str(tmp_path), 300, barrier, q)) for _ in range(4)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=120) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all(r[1] and r[2] and r[3] == 300 for r in res), res
    packs = list((tmp_path / "embeddings" / "packs").glob("pack-*"))
    assert sorted(q_.suffix for q_ in packs) == [".json", ".npy"], packs                      # one pack, no temporary files left behind


def test_window_start_tables_reproduce_the_host_cut_windows():
    """The ingest path ships a recording + int32 start tables instead of materialised windows: the tables must describe exactly the windows
    cut_windows / cut_ranges produce (zero padding past the end included)."""
    rng = np.random.default_rng(5)
    x = rng.integers(-3000, 3000, 16000 * 9 + 777).astype(np.int16)
    for segs in (None, [(1.0, 1.2), (3.0, 4.0), (5.0, 9.5)], [(0.0, 0.7)]):
        st, spans, W = wav.window_starts(len(x), segs)
        pcm, spans2 = wav.cut_windows(x, segs)
        assert st.dtype == np.int32 and spans == spans2 and np.array_equal(wav.materialise_windows(x, st, W), pcm)
        assert ((st >= 0) & (st < len(x))).all()
    short = x[:20000]                                            # shorter than a window: one zero-padded window starting at 0
    st, spans, W = wav.window_starts(len(short), None)
    assert st.tolist() == [0] and wav.materialise_windows(short, st, W)[0, 20000:].max() == 0
    ranges = [(0.2, 3.1), (3.3, 4.0), (4.2, 9.0), (9.1, 9.3)]
    tabs, wins, dropped = wav.range_starts(len(x), ranges)
    pcm_by_len, wins2, dropped2 = wav.cut_ranges(x, ranges)
    assert wins == wins2 and dropped == dropped2 == [3] and set(tabs) == set(pcm_by_len)
    for S, st in tabs.items():
        assert np.array_equal(wav.materialise_windows(x, st, S), pcm_by_len[S]) and (st + S <= len(x)).all()


def test_vectors_remember_their_numerical_setting(tmp_path, monkeypatch):
    """model_version names the weights, not SDK_BIAS_CORRECTION / SDK_PRECISION; a vector enrolled under another setting is
    comparable (same space, ~4e-3 apart) - it is kept and WARNED about, never silently mixed; the pack is per setting."""
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    rng = np.random.default_rng(2)
    made = {"precision": 0, "bias_correction": True}
    cands = []
    for i in range(20):
        v = rng.standard_normal(192).astype(np.float32)
        ext = store.save_vector(v, meta=made if i % 2 == 0 else None)          # odd ones: enrolled before the sidecar existed - no claim, no warning
        cands.append(_profile(f"s{i}", [{"id": f"e{i}", "external_id": ext, "model_version": "mi355x-ecapa1024-x"}]))
    assert store.load_vector_meta(cands[0]["embeddings"]["mi355x"][0]["external_id"]) == made
    assert store.load_vector_meta(cands[1]["embeddings"]["mi355x"][0]["external_id"]) is None
    kw = dict(model_prefix="mi355x-", model_version="mi355x-ecapa1024-x")
    same = store.load_profile_batch(cands, "mi355x", settings=made, **kw)
    assert len(same) == 20 and same.warnings == []
    other = store.load_profile_batch(cands, "mi355x", settings={"precision": 0, "bias_correction": False}, **kw)
    assert len(other) == 20 and len(other.warnings) == 10 and "bias_correction" in other.warnings[0]
    assert other.pack_ref[2] != same.pack_ref[2]                                # another setting, another pack
    store.publish_pack(other, *_fake_norm(other.matrix))
    again = store.load_profile_batch(cands, "mi355x", settings={"precision": 0, "bias_correction": False}, **kw)
    assert again.from_pack and again.warnings == other.warnings                  # the warnings travel with the pack
    assert not store.load_profile_batch(cands, "mi355x", settings=made, **kw).from_pack
    be = backend.Backend()
    monkeypatch.setenv("SDK_PRECISION", "1")
    assert be.numerics() == {"precision": 1, "bias_correction": False}


def test_db_listing_pack_follows_every_change(tmp_path, monkeypatch, capsys):
    """a7 at BASELINE's profile counts: the in-process counterpart of `speaker_detection identify` (which reads every db/*.json per call,
    speaker_detection:206-220) keeps the parsed list as one file keyed by the directory listing; an edit, an addition and a deletion each
    invalidate it; a broken file warns on a miss AND on a hit.  The pack lives OUTSIDE db/ (ADVICE r4 high): db/ is shared with the reference
    CLI, whose `db_path.glob("*.json")` matches dot-files - listed exactly that way below, db/ must hold the real profiles and nothing else."""
    ident = sub("identify")
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path))
    db = tmp_path / "db"
    db.mkdir()
    for i in range(70):
        (db / f"s{i:03d}.json").write_text(json.dumps({"id": f"s{i:03d}", "tags": ["a"] if i % 2 else [], "embeddings": {"mi355x": [{"id": f"e{i}"}]}}))
    (db / "broken.json").write_text("{not json")
    first = ident.list_all_speakers()
    pack = tmp_path / "cache" / "profiles-pack.json"
    assert [p["id"] for p in first] == [f"s{i:03d}" for i in range(70)] and pack.exists() and ident.listing_pack_path() == pack
    # the reference's own listing of db/ after the pack was built (speaker_detection:213): the 70 profiles + the broken file, every one with an 'id'
    ref_listing = sorted(db.glob("*.json"))
    assert [q.name for q in ref_listing] == ["broken.json"] + [f"s{i:03d}.json" for i in range(70)]
    assert all("id" in json.loads(q.read_text()) for q in ref_listing if q.name != "broken.json")
    assert [q.name for q in db.iterdir() if q.name.startswith(".")] == []          # no temp / cache files left inside db/ either
    assert "broken.json" in capsys.readouterr().err
    reads = []
    real = Path.read_text
    monkeypatch.setattr(Path, "read_text", lambda self, *a, **k: (reads.append(self.name), real(self, *a, **k))[1])
    again = ident.list_all_speakers()
    assert again == first and reads == ["profiles-pack.json"]                      # ONE file opened, whatever the number of speakers
    assert "broken.json" in capsys.readouterr().err
    (db / "s005.json").write_text(json.dumps({"id": "s005", "tags": ["edited"], "embeddings": {}}))       # edit (size and mtime change)
    assert ident.list_all_speakers()[5]["tags"] == ["edited"]
    # a same-size rewrite with the old mtime restored (coarse-mtime file systems, mtime-preserving tools): ctime still moves
    st = (db / "s006.json").stat()
    body = (db / "s006.json").read_text()
    (db / "s006.json").write_text(body.replace('"e6"', '"E6"'))
    os.utime(db / "s006.json", ns=(st.st_atime_ns, st.st_mtime_ns))
    assert (db / "s006.json").stat().st_size == st.st_size and (db / "s006.json").stat().st_mtime_ns == st.st_mtime_ns
    assert ident.list_all_speakers()[6]["embeddings"]["mi355x"][0]["id"] == "E6"
    (db / "s070.json").write_text(json.dumps({"id": "s070", "embeddings": {}}))                           # addition
    assert len(ident.list_all_speakers()) == 71
    (db / "s000.json").unlink()                                                                           # deletion
    assert ident.list_all_speakers()[0]["id"] == "s001" and len(ident.list_all_speakers()) == 70
    # a dot-file in db/ IS a profile to the reference's glob, so it is one here too (and sorts first)
    (db / ".hidden.json").write_text(json.dumps({"id": "hidden", "embeddings": {}}))
    assert ident.list_all_speakers()[0]["id"] == "hidden" and len(ident.list_all_speakers()) == 71
    (db / ".hidden.json").unlink()
    assert len(ident.candidates_for("mi355x", tags=["a"])) == 34
    monkeypatch.setenv("SDK_PROFILE_PACK", "0")
    reads.clear()
    assert len(ident.list_all_speakers()) == 70 and "profiles-pack.json" not in reads and len(reads) == 71
