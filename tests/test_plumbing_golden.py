"""Host-side plumbing vs golden vectors captured from the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import json
import struct
import random
from pathlib import Path

import pytest

from conftest import sub

ac = sub("audio_contract")
seg = sub("segments")
asg = sub("assign")
api = sub("plugin_api")


def _norm(x):
    """JSON round-trip (tuples -> lists) so structures compare like the golden file's."""
    return json.loads(json.dumps(x))


def test_audio_profiles_named(golden):
    for name, rec in golden["audio_profiles"]["named"].items():
        p = ac.get_profile(name)
        assert [p.sample_rate, p.channels, p.format, p.bit_depth, p.max_duration_sec] == rec["fields"]
        assert ac.format_ffmpeg_args(p) == rec["ffmpeg"]
    assert ac.format_ffmpeg_args(ac.get_profile("mi355x")) == ["-ar", "16000", "-ac", "1", "-f", "wav", "-acodec", "pcm_s16le"]


def test_audio_profiles_variants(golden):
    for v in golden["audio_profiles"]["variants"]:
        sr, ch, fmt, bd = v["args"]
        assert ac.format_ffmpeg_args(ac.AudioProfile(sample_rate=sr, channels=ch, format=fmt, bit_depth=bd)) == v["ffmpeg"]


def test_get_profile_unknown_and_register(golden):
    """test_audio_profiles.py:83-137 - an unknown backend gets the default profile; a registered one is returned as is."""
    d, u = ac.get_profile("default"), ac.get_profile("nonexistent_backend")
    assert (u.sample_rate, u.channels, u.format, u.bit_depth) == (d.sample_rate, d.channels, d.format, d.bit_depth)
    custom = ac.AudioProfile(sample_rate=44100, channels=2, format="mp3", bit_depth=24, max_duration_sec=300.0)
    ac.register_profile("custom_test", custom)
    try:
        got = ac.get_profile("custom_test")
        assert (got.sample_rate, got.channels, got.format, got.bit_depth, got.max_duration_sec) == (44100, 2, "mp3", 24, 300.0)
        assert ac.format_ffmpeg_args(got) == ["-ar", "44100", "-ac", "2", "-f", "mp3"]        # no -acodec outside wav
    finally:
        ac.PROFILES.pop("custom_test", None)


def _transcripts(golden, fixture_transcript_path):
    cases = dict(golden["transcript"]["inputs"])
    cases["fixture"] = json.loads(Path(fixture_transcript_path).read_text())
    return cases


def test_transcript_parsers(golden, fixture_transcript_path):
    cases = _transcripts(golden, fixture_transcript_path)
    for cname, want in golden["transcript"]["outputs"].items():
        data = cases[cname]
        assert seg.detect_transcript_format(data) == want["format_backend"], cname
        assert asg.detect_transcript_format(data) == want["format_assign"], cname
        assert seg.get_available_speakers(data) == want["speakers_backend"], cname
        assert asg.get_speakers_from_transcript(data) == want["speakers_assign"], cname
        for lab, rec in want["labels"].items():
            assert _norm(seg.extract_segments_as_tuples(data, lab)) == rec["tuples"], (cname, lab)
            assert _norm(seg.extract_segments_from_transcript(data, lab)) == rec["merged_default"], (cname, lab)
            assert _norm(seg.extract_segments_from_transcript(data, lab, min_duration=0.1, max_gap=0.3)) == rec["merged_0.1_0.3"], (cname, lab)
            assert _norm(asg.get_speaker_segments(data, lab)) == rec["assign_segments"], (cname, lab)


def test_fixture_known_values(fixture_transcript_path):
    data = seg.load_transcript(fixture_transcript_path)
    assert seg.extract_segments_as_tuples(data, "Alice") == [(0.04, 5.36)]
    assert seg.extract_segments_as_tuples(data, "Bob") == [(5.36, 11.44)]
    sent = seg.sentence_segments(data)
    assert [(s["start"], s["end"]) for s in sent if s["speaker"] == "Alice"] == [(0.04, 0.88), (0.92, 1.92), (2.0, 3.32), (3.36, 4.6), (4.64, 5.36)]
    assert [(s["start"], s["end"]) for s in sent if s["speaker"] == "Bob"] == [(5.36, 6.76), (6.8, 7.28), (7.28, 9.08), (9.12, 9.72), (9.76, 11.44)]
    assert len(sent) == 10


def test_merge_segments_by_gap(golden):
    for c in golden["merge_segments_by_gap"]:
        got = seg.merge_segments_by_gap([tuple(x) for x in c["segments"]], c["max_gap"])
        assert _norm(got) == c["out"]


def test_constants(golden):
    k = golden["constants"]
    assert asg.SIGNAL_WEIGHTS == k["SIGNAL_WEIGHTS"] and asg.TRUST_MULTIPLIERS == k["TRUST_MULTIPLIERS"]
    assert asg.CONFIDENCE_THRESHOLDS == k["CONFIDENCE_THRESHOLDS"]
    assert asg.VERSION == k["VERSION"] and asg.SCHEMA_VERSION == k["SCHEMA_VERSION"]


def test_combine_signals_bit_exact(golden):
    """float64 scores must match to the last bit (order-dependent accumulation, stable sort)."""
    assert len(golden["combine_signals"]) >= 50
    for kat in golden["combine_signals"]:
        sigs = [asg.Signal(type=t, speaker_id=sid, score=sc, evidence=dict(ev)) for t, sid, sc, ev in kat["signals"]]
        a = asg.combine_signals(kat["label"], sigs, threshold=kat["threshold"])
        want = kat["out"]
        assert a.speaker_id == want["speaker_id"], kat["name"]
        assert a.confidence == want["confidence"], kat["name"]
        assert float(a.score).hex() == want["score_hex"], kat["name"]
        assert a.signals == want["signals"], kat["name"]
        assert [c["speaker_id"] for c in a.candidates] == [c["speaker_id"] for c in want["candidates"]], kat["name"]
        assert [float(c["score"]).hex() for c in a.candidates] == [float(c["score"]).hex() for c in want["candidates"]], kat["name"]


def test_known_answers_from_survey():
    E = lambda t: {"trust_level": t}
    a = asg.combine_signals("S1", [asg.Signal("embedding_match", "alice", .9, E("high")), asg.Signal("embedding_match", "bob", .7, E("high"))], .3)
    assert (a.speaker_id, a.confidence, repr(a.score)) == ("alice", "low", "0.36000000000000004")
    assert repr(a.candidates[0]["score"]) == "0.27999999999999997"
    a = asg.combine_signals("S1", [asg.Signal("embedding_match", "alice", .8, E("low")), asg.Signal("embedding_match", "bob", .5, E("high"))], .1)
    assert (a.speaker_id, repr(a.score), repr(a.candidates[0]["score"])) == ("bob", "0.2", "0.12800000000000003")   # DEV_NOTES.md:351-376


def test_compute_trust_level(golden):
    for c in golden["compute_trust_level"]:
        assert asg.compute_trust_level(c["samples"]) == c["out"]


class _Probe(api._MirrorBackend):
    @property
    def name(self): return "mi355x"
    @property
    def requires_api_key(self): return False
    def enroll_speaker(self, audio_path, segments=None): return {}
    def identify_speaker(self, audio_path, candidates, threshold=0.354):
        return [{"speaker_id": c["id"], "similarity": 0.5, "embedding_id": "emb-1"} for c in candidates]


def test_abc_defaults(golden, fixture_transcript_path):
    g = golden["abc_defaults"]
    p = _Probe()
    assert p.embedding_dim == g["embedding_dim"] and p.model_version == g["model_version"] and p.audio_profile == g["audio_profile"]
    assert p.check_embedding_compatibility({"model_version": "mi355x-ecapa-1"}) == g["compat_ok"]
    assert p.check_embedding_compatibility({"model_version": "speechmatics-v2"}) == g["compat_bad"]
    assert p.check_embedding_compatibility({}) == g["compat_missing"]
    assert p.verify_speaker(Path("x.wav"), {"id": "alice"}) == g["verify_hit"]
    assert _norm(p.extract_segments_from_transcript(fixture_transcript_path, "Alice")) == g["segments_from_transcript_Alice"]
    with pytest.raises(TypeError):
        api._MirrorBackend()          # abstract, like the reference ABC


def _write_wav(path, seconds=1.0, seed=7):
    rng = random.Random(seed)
    n = int(16000 * seconds)
    pcm = b"".join(struct.pack("<h", rng.randint(-3000, 3000)) for _ in range(n))
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" + struct.pack("<I", len(pcm))
    Path(path).write_bytes(hdr + pcm)


def _args_to_kwargs(argv):
    kw = {}
    it = iter(argv)
    for a in it:
        if a == "--use-embeddings": kw["use_embeddings"] = True
        elif a == "--min-trust": kw["min_trust"] = next(it)
        elif a == "--threshold": kw["threshold"] = float(next(it))
        elif a == "--expected-speakers": kw["expected_speakers"] = next(it).split(",")
        elif a == "--context": kw["context"] = next(it)
        elif a == "--format": next(it)
    return kw


def test_cmd_assign_outputs(golden, fixture_transcript_path, tmp_path, monkeypatch):
    """Config #1 of BASELINE.json: the reference CLI's JSON (and saved YAML) for the committed
    fixture, reproduced by assign_recording with the same identify rows."""
    monkeypatch.setenv("PATH", "/usr/bin:/bin")          # no b3sum -> sha256 fallback, as in the capture
    wav = tmp_path / "a.wav"
    _write_wav(wav, seed=golden["cli"]["wav_seed"])
    assert asg.compute_b3sum(wav) == golden["cli"]["wav_sha256_32"]
    for run in golden["cli"]["runs"]:
        rows = run["stub_rows"]
        kw = _args_to_kwargs(run["argv"])
        out = asg.assign_recording(wav, fixture_transcript_path, rows_fn=(lambda label, segs: rows) if rows is not None else None, **kw)
        out["transcript_path"] = "<TRANSCRIPT>"
        out.pop("assigned_at")
        assert _norm(out) == run["json"], run["name"]
        if "saved_yaml_obj" in run:
            import yaml
            full = asg.assign_recording(wav, fixture_transcript_path, rows_fn=lambda label, segs: rows, **kw)
            path = asg.save_assignment(full, tmp_path / "saved.yaml")
            y = yaml.safe_load(path.read_text())
            y["transcript_path"] = "<TRANSCRIPT>"
            y.pop("assigned_at")
            assert y == run["saved_yaml_obj"]


def test_rows_with_trust_shape():
    profs = {"alice": {"id": "alice", "names": {"default": "Alice"}, "embeddings": {"mi355x": [
        {"id": "emb-1", "trust_level": "low"}, {"id": "emb-2", "trust_level": "high"}]}}}
    rows = asg.rows_with_trust([{"speaker_id": "alice", "similarity": 0.5}, {"speaker_id": "zed", "confidence": 0.4, "embedding_id": "emb-9"}], profs, "mi355x")
    assert list(rows[0].keys()) == ["speaker_id", "name", "score", "confidence", "trust_level", "embedding_id", "backend"]  # speaker_detection:1115-1123
    assert rows[0]["trust_level"] == "high" and rows[0]["embedding_id"] == "emb-2" and rows[0]["name"] == "Alice"
    assert rows[1]["name"] == "zed" and rows[1]["trust_level"] == "unknown" and rows[1]["score"] == 0.4
