#!/usr/bin/env python3
"""Generate golden vectors for the *plumbing* rows of SURVEY.md §8 (a5, a9-a12).

Runs ONLY in the build container, where the upstream reference is mounted at
/root/reference.  It imports / executes the reference's own Python and records
inputs + outputs as JSON data.  Nothing from the reference travels: the outputs
are plain data (tests/golden/*.json), and this script is the committed recipe
that made them.  The GPU box never runs this file.

    python3 tests/golden/make_golden.py          # rewrites tests/golden/*.json

Reference entry points exercised (file:line):
  speaker-assign:418-492      combine_signals
  speaker-assign:169-246      detect_transcript_format / get_speakers_from_transcript /
                              get_speaker_segments
  speaker-assign:262-328      collect_embedding_signals  (through a canned
                              `speaker_detection` stub on PATH)
  speaker-assign:499-649      cmd_assign (CLI, --dry-run --format json and saved YAML)
  speaker_detection_backends/transcript.py:25-305
  speaker_detection_backends/audio_profiles.py:12-100
  speaker_detection_backends/base.py:73-105,153-180   compat check / verify default
  speaker_detection:359-379   compute_trust_level
  speaker_segments:38-71      merge_segments_by_gap
"""
from __future__ import annotations

import importlib.machinery
import importlib.util
import json
import os
import random
import shutil
import stat
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent
FIXTURE_SRC = REF / "evals/speaker_detection/audio/test_001-two-speakers.wav.speechmatics.json"
FIXTURE_DST = HERE / "test_001-two-speakers.wav.speechmatics.json"


def load_script(name: str, path: Path):
    loader = importlib.machinery.SourceFileLoader(name, str(path))
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    loader.exec_module(mod)
    return mod


def write_wav(path: Path, seconds: float = 1.0, seed: int = 7) -> None:
    """1-s 16 kHz mono s16le WAV, seeded, byte-stable (so sha256[:32] is stable)."""
    rng = random.Random(seed)
    n = int(16000 * seconds)
    pcm = b"".join(struct.pack("<h", rng.randint(-3000, 3000)) for _ in range(n))
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack(
        "<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" + struct.pack("<I", len(pcm))
    path.write_bytes(hdr + pcm)


def make_schema_golden(ref: Path, here: Path) -> None:
    """(1) verdicts of the reference's own validators (schemas.py:45-251) on the records a local backend
    produces - good ones, and ones with missing keys / wrong types; (2) the embedding record the reference's `enroll` CLI
    (speaker_detection:754-919) writes when a registered backend returns `external_id = "npy:<key>"`, run black-box with a
    canned backend module registered through $SPEAKER_BACKENDS_CONFIG."""
    from speaker_detection_backends import schemas as rs

    def verdict(fn, obj):
        v = {"warnings": fn(obj, strict=False)}
        try:
            fn(obj, strict=True)
            v["strict_error"] = None
        except rs.ValidationError as exc:
            v["strict_error"] = str(exc)
        return v

    good = {"id": "emb-0a1b2c3d", "external_id": "npy:0123456789abcdef01234567", "source_audio": "/data/rec/a.wav",
            "source_audio_b3sum": "ab" * 16, "source_segments": [{"start": 0.04, "end": 5.36}],
            "model_version": "mi355x-ecapa1024-2f6c1e0d9b7a", "samples": {"reviewed": [], "unreviewed": [], "rejected": []},
            "trust_level": "low", "created_at": "2026-01-13T10:11:12.131415+00:00"}

    def variant(**kw):
        r = dict(good)
        for k, v in kw.items():
            if v == "<drop>":
                r.pop(k)
            else:
                r[k] = v
        return r

    emb_cases = [
        ("good", good),
        ("good_no_segments_z_time", variant(source_segments=None, created_at="2026-01-13T10:11:12Z")),
        ("with_all_identifiers", variant(all_identifiers=["npy:0123456789abcdef01234567"])),
        ("external_id_null", variant(external_id=None)),
        ("external_id_int", variant(external_id=17)),
        ("missing_external_id", variant(external_id="<drop>")),
        ("missing_created_and_id", variant(created_at="<drop>", id="<drop>")),
        ("empty_id", variant(id="")),
        ("model_unknown", variant(model_version="unknown")),
        ("model_not_str", variant(model_version=3)),
        ("trust_bogus", variant(trust_level="certain")),
        ("trust_invalidated", variant(trust_level="invalidated")),
        ("created_not_iso", variant(created_at="yesterday")),
        ("created_not_str", variant(created_at=1736762400)),
        ("samples_bad_types", variant(samples={"reviewed": "abc", "unreviewed": [1, 2], "rejected": []})),
        ("samples_list", variant(samples=["x"])),
        ("samples_null", variant(samples=None)),
        ("segments_not_list", variant(source_segments="0-5")),
        ("segments_malformed", variant(source_segments=[{"start": 1.0}, [2.0, 3.0], {"start": 4.0, "end": 5.0}])),
        ("not_a_dict", ["emb-1"]),
    ]
    prof_good = {"id": "alice", "version": 1, "names": {"default": "Alice"}, "nicknames": [], "description": None,
                 "metadata": {}, "tags": ["team"], "embeddings": {"mi355x": [good]},
                 "created_at": "2026-01-13T10:00:00+00:00", "updated_at": "2026-01-13T10:11:12+00:00"}

    def pvariant(**kw):
        r = json.loads(json.dumps(prof_good))
        for k, v in kw.items():
            if v == "<drop>":
                r.pop(k)
            else:
                r[k] = v
        return r

    prof_cases = [
        ("good", prof_good),
        ("no_default_name", pvariant(names={"work": "A."})),
        ("names_list", pvariant(names=["Alice"])),
        ("missing_names", pvariant(names="<drop>")),
        ("empty_id", pvariant(id="")),
        ("tags_str", pvariant(tags="team")),
        ("tags_mixed", pvariant(tags=["a", 1])),
        ("embeddings_list", pvariant(embeddings=[good])),
        ("backend_not_list", pvariant(embeddings={"mi355x": good})),
        ("nested_bad_record", pvariant(embeddings={"mi355x": [good, variant(external_id=5, trust_level="x"), "oops"]})),
        ("version_str", pvariant(version="1")),
        ("not_a_dict", "alice"),
    ]
    out = {"embedding": [{"name": n, "record": r, **verdict(rs.validate_embedding, r)} for n, r in emb_cases],
           "profile": [{"name": n, "profile": p, **verdict(rs.validate_profile, p)} for n, p in prof_cases]}

    # ---- the record the reference's enroll CLI writes for a registered local backend ---------------------------------
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        (td / "cannedbackend.py").write_text(
            "from pathlib import Path\n"
            "from speaker_detection_backends.base import EmbeddingBackend\n"
            "class Backend(EmbeddingBackend):\n"
            "    @property\n"
            "    def name(self): return 'mi355x'\n"
            "    @property\n"
            "    def requires_api_key(self): return False\n"
            "    @property\n"
            "    def model_version(self): return 'mi355x-ecapa1024-2f6c1e0d9b7a'\n"
            "    def enroll_speaker(self, audio_path, segments=None):\n"
            "        return {'external_id': 'npy:0123456789abcdef01234567', 'file': 'dropped.npy', 'model_version': self.model_version,\n"
            "                'source_audio': str(audio_path), 'source_segments': segments}\n"
            "    def identify_speaker(self, audio_path, candidates, threshold=0.354): return []\n")
        (td / "backends.yaml").write_text("backends:\n  mi355x:\n    module: cannedbackend\n")
        wav = td / "a.wav"
        write_wav(wav)
        env = dict(os.environ)
        env.update({"SPEAKERS_EMBEDDINGS_DIR": str(td / "store"), "SPEAKER_BACKENDS_CONFIG": str(td / "backends.yaml"),
                    "PYTHONPATH": f"{td}:{ref}", "PYTHONDONTWRITEBYTECODE": "1", "PATH": "/usr/bin:/bin"})
        cli = [sys.executable, str(ref / "speaker_detection")]
        runs = []
        assert subprocess.run(cli + ["add", "alice", "--name", "Alice"], env=env, capture_output=True, text=True).returncode == 0
        for name, extra in (("whole_file", []), ("segments", ["--segments", "0.04:0.5,0.6:0.9"]),
                            ("trust_override", ["--trust-level", "high"])):
            r = subprocess.run(cli + ["enroll", "alice", str(wav), "-b", "mi355x"] + extra, env=env, capture_output=True, text=True)
            prof = json.loads((td / "store" / "db" / "alice.json").read_text())
            rec = prof["embeddings"]["mi355x"][-1]
            runs.append({"name": name, "argv": extra, "rc": r.returncode, "record": rec,
                         "validate": rs.validate_embedding(rec), "profile_validate": rs.validate_profile(prof)})
        import hashlib
        out["enroll_cli"] = {"wav_seed": 7, "wav_sha256_32": hashlib.sha256(wav.read_bytes()).hexdigest()[:32],
                             "audio_path": str(wav), "runs": runs}
    (here / "schema_golden.json").write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", here / "schema_golden.json")


def main() -> int:
    if not REF.exists():
        print("reference not mounted; golden vectors can only be regenerated in the build container",
              file=sys.stderr)
        return 2
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    from speaker_detection_backends import transcript as rt
    from speaker_detection_backends import audio_profiles as rap
    from speaker_detection_backends import base as rbase

    sa = load_script("ref_speaker_assign", REF / "speaker-assign")
    sd = load_script("ref_speaker_detection", REF / "speaker_detection")
    sseg = load_script("ref_speaker_segments", REF / "speaker_segments")

    shutil.copyfile(FIXTURE_SRC, FIXTURE_DST)  # data file held by the reference's tests
    fixture = json.loads(FIXTURE_SRC.read_text())

    out: dict = {}

    # ---- a5: audio profile contract -------------------------------------------------
    prof = {}
    for name in ["speechmatics", "pyannote", "default", "no-such-backend"]:
        p = rap.get_profile(name)
        prof[name] = {"fields": [p.sample_rate, p.channels, p.format, p.bit_depth, p.max_duration_sec],
                      "ffmpeg": rap.format_ffmpeg_args(p)}
    variants = []
    for fmt, bd, sr, ch in [("wav", 16, 16000, 1), ("wav", 24, 8000, 2), ("wav", 32, 44100, 1),
                            ("wav", 8, 16000, 1), ("wav", 12, 16000, 1), ("mp3", 16, 22050, 2)]:
        variants.append({"args": [sr, ch, fmt, bd],
                         "ffmpeg": rap.format_ffmpeg_args(rap.AudioProfile(sample_rate=sr, channels=ch, format=fmt, bit_depth=bd))})
    out["audio_profiles"] = {"named": prof, "variants": variants}

    # ---- a12 + speaker-assign parsers on the committed fixture and synthetic ones ----
    aai = {"utterances": [
        {"speaker": "A", "start": 0, "end": 1500, "text": "hello there"},
        {"speaker": "B", "start": 1600, "end": 2100, "text": "hi"},
        {"speaker": "A", "start": 2200, "end": 2500, "text": "ok"},
        {"speaker": "A", "start": 2900, "end": 5250, "text": "let us begin"},
        {"speaker": "B", "start": 9000, "end": 9400, "text": "sure"},
    ]}
    sm_top = {"results": [  # speechmatics, speaker on the item (no identification)
        {"type": "word", "start_time": 0.0, "end_time": 0.4, "speaker": "S1", "alternatives": [{"content": "a"}]},
        {"type": "word", "start_time": 0.5, "end_time": 0.9, "speaker": "S1", "alternatives": [{"content": "b"}]},
        {"type": "punctuation", "start_time": 0.9, "end_time": 0.9, "is_eos": True, "alternatives": [{"content": "."}]},
        {"type": "word", "start_time": 1.0, "end_time": 1.3, "speaker": "S2", "alternatives": [{"content": "c"}]},
        {"type": "word", "start_time": 1.4, "end_time": 2.6, "alternatives": [{"content": "d"}]},
        {"type": "word", "start_time": 2.7, "end_time": 3.9, "speaker": "S1", "alternatives": [{"content": "e"}]},
        {"type": "word", "start_time": 4.0, "end_time": 4.1, "speaker": "S1", "alternatives": [{"content": "f"}]},
        {"type": "word", "start_time": 8.0, "end_time": 8.9, "speaker": "S1", "alternatives": [{"content": "g"}]},
    ]}
    tcases = {"fixture": fixture, "assemblyai": aai, "speechmatics_top": sm_top,
              "empty_results": {"results": []}, "unknown": {"foo": 1}}
    tout = {}
    for cname, data in tcases.items():
        labels = sorted(set(rt.get_available_speakers(data)) | set(sa.get_speakers_from_transcript(data)) | {"UU", "nobody"})
        rec = {
            "format_backend": rt.detect_transcript_format(data),
            "format_assign": sa.detect_transcript_format(data),
            "speakers_backend": rt.get_available_speakers(data),
            "speakers_assign": sa.get_speakers_from_transcript(data),
            "labels": {},
        }
        for lab in labels:
            rec["labels"][lab] = {
                "tuples": rt.extract_segments_as_tuples(data, lab),
                "merged_default": rt.extract_segments_from_transcript(data, lab),
                "merged_0.1_0.3": rt.extract_segments_from_transcript(data, lab, min_duration=0.1, max_gap=0.3),
                "assign_segments": sa.get_speaker_segments(data, lab),
            }
        tout[cname] = rec
    out["transcript"] = {"inputs": {k: v for k, v in tcases.items() if k != "fixture"}, "outputs": tout}

    # speaker_segments:38-71
    mcases = [([], 1.0), ([(0.0, 1.0)], 1.0), ([(0.0, 1.0), (1.5, 2.0), (4.0, 5.0)], 1.0),
              ([(0.0, 1.0), (1.5, 2.0), (4.0, 5.0)], 0.0), ([(0.0, 1.0), (2.0, 3.0), (3.0, 3.5), (9.0, 9.5)], 1.0),
              ([(0.0, 1.0), (1.5, 2.0)], -1.0)]
    out["merge_segments_by_gap"] = [{"segments": s, "max_gap": g,
                                     "out": sseg.merge_segments_by_gap([tuple(x) for x in s], g)} for s, g in mcases]

    # ---- a10: combine_signals known answers (float64 exact, via repr round-trip) -----
    def run_combine(label, sigs, thr):
        signals = [sa.Signal(type=t, speaker_id=sid, score=sc, evidence=dict(ev)) for t, sid, sc, ev in sigs]
        a = sa.combine_signals(label, signals, threshold=thr)
        return {"speaker_id": a.speaker_id, "confidence": a.confidence, "score": a.score,
                "score_hex": float(a.score).hex(), "signals": a.signals, "candidates": a.candidates}

    E = lambda t: {"trust_level": t}
    kats = [
        ("kat1", [("embedding_match", "alice", .9, E("high")), ("embedding_match", "bob", .7, E("high"))], .3),
        ("kat2", [("embedding_match", "alice", .8, E("low")), ("embedding_match", "bob", .5, E("high"))], .1),
        ("kat3", [("embedding_match", "alice", .85, E("high")), ("context_expected", "alice", .5, {"context": None, "reason": "x"}),
                  ("llm_name_detection", "alice", .9, {"detected_name": "Alice", "evidence": []}),
                  ("context_expected", "bob", .5, {"context": None, "reason": "x"})], .3),
        ("kat4", [("embedding_match", "alice", .99, E("invalidated"))], .3),
        ("kat5", [("context_expected", "bob", .5, {}), ("context_expected", "alice", .5, {})], .05),
        ("kat6", [("weird", "zed", 1.0, {})], .05),
        ("kat7", [("embedding_match", "al", 1.0, {})], .1),
        ("kat8_none_id", [("embedding_match", None, .9, E("high"))], .3),
        ("kat9_empty", [], .5),
        ("kat10_five", [("embedding_match", s, sc, E(t)) for s, sc, t in
                        [("a", .91, "high"), ("b", .9, "high"), ("c", .89, "medium"), ("d", .6, "low"), ("e", .95, "unknown")]], .2),
        ("kat11_bands", [("embedding_match", "a", 1.0, E("high")), ("llm_name_detection", "a", 1.0, {})], .3),
        ("kat12_band_medium", [("embedding_match", "a", 1.0, E("high"))], .3),
        ("kat13_dup_ids", [("embedding_match", "a", .5, E("high")), ("embedding_match", "a", .4, E("medium")),
                           ("embedding_match", "b", .7, E("medium"))], .1),
    ]
    rng = random.Random(20260101)
    types = ["embedding_match", "llm_name_detection", "context_expected", "cross_backend_agreement", "other"]
    trusts = ["high", "medium", "low", "invalidated", "unknown", "bogus", None]
    for i in range(40):
        n = rng.randint(1, 12)
        sigs = []
        for _ in range(n):
            t = rng.choice(types)
            ev = {}
            tr = rng.choice(trusts)
            if t == "embedding_match" and tr is not None:
                ev = {"trust_level": tr, "embedding_id": "emb-%08x" % rng.getrandbits(32), "backend": "mi355x"}
            sigs.append((t, rng.choice(["alice", "bob", "carol", "dave", None]), round(rng.random(), rng.choice([2, 6, 17])), ev))
        kats.append((f"rand{i:02d}", sigs, rng.choice([0.05, 0.1, 0.3, 0.5])))
    out["combine_signals"] = [{"name": n, "label": "S1", "threshold": thr,
                               "signals": [list(s) for s in sigs], "out": run_combine("S1", sigs, thr)}
                              for n, sigs, thr in kats]
    out["constants"] = {"SIGNAL_WEIGHTS": sa.SIGNAL_WEIGHTS, "TRUST_MULTIPLIERS": sa.TRUST_MULTIPLIERS,
                        "CONFIDENCE_THRESHOLDS": sa.CONFIDENCE_THRESHOLDS, "VERSION": sa.VERSION,
                        "SCHEMA_VERSION": sa.SCHEMA_VERSION}

    # ---- speaker_detection:359-379 compute_trust_level --------------------------------
    tl = []
    for r, u, x in [([], [], []), (["a"], [], []), (["a"], ["b"], []), ([], ["b"], []), (["a"], [], ["c"]), ([], [], ["c"])]:
        tl.append({"samples": {"reviewed": r, "unreviewed": u, "rejected": x},
                   "out": sd.compute_trust_level({"reviewed": r, "unreviewed": u, "rejected": x})})
    out["compute_trust_level"] = tl

    # ---- base.py defaults for a minimal subclass (a3/a4) --------------------------------
    class Probe(rbase.EmbeddingBackend):
        @property
        def name(self): return "mi355x"
        @property
        def requires_api_key(self): return False
        def enroll_speaker(self, audio_path, segments=None): return {}
        def identify_speaker(self, audio_path, candidates, threshold=0.354):
            return [{"speaker_id": c["id"], "similarity": 0.5, "embedding_id": "emb-1"} for c in candidates]
    pb = Probe()
    out["abc_defaults"] = {
        "embedding_dim": pb.embedding_dim, "model_version": pb.model_version, "audio_profile": pb.audio_profile,
        "compat_ok": pb.check_embedding_compatibility({"model_version": "mi355x-ecapa-1"}),
        "compat_bad": pb.check_embedding_compatibility({"model_version": "speechmatics-v2"}),
        "compat_missing": pb.check_embedding_compatibility({}),
        "verify_hit": pb.verify_speaker(Path("x.wav"), {"id": "alice"}),
        "list_backends_default": rbase.list_backends(),
        "segments_from_transcript_Alice": pb.extract_segments_from_transcript(FIXTURE_SRC, "Alice"),
    }

    # ---- a9/a11: the CLI itself, black-box, with a canned `speaker_detection` on PATH ----
    cli = []
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        wav = td / "a.wav"
        write_wav(wav)
        stub_dir = td / "bin"
        stub_dir.mkdir()
        rows = [
            {"speaker_id": "alice", "name": "Alice", "score": 0.8123456, "confidence": 0.8123456,
             "trust_level": "high", "embedding_id": "emb-aaaa0001", "backend": "mi355x"},
            {"speaker_id": "bob", "name": "Bob", "score": 0.6400001, "confidence": 0.6400001,
             "trust_level": "medium", "embedding_id": "emb-bbbb0002", "backend": "mi355x"},
            {"speaker_id": "carol", "name": "Carol", "score": 0.99, "confidence": 0.99,
             "trust_level": "low", "embedding_id": "emb-cccc0003", "backend": "mi355x"},
            {"speaker_id": "dave", "name": "Dave", "score": 0.7, "confidence": 0.7,
             "trust_level": "unknown", "embedding_id": None, "backend": "mi355x"},
            {"name": "nobody", "score": 0.9},
        ]
        stub = stub_dir / "speaker_detection"
        stub.write_text("#!/bin/sh\ncat <<'EOF'\n" + json.dumps(rows, indent=2) + "\nEOF\n")
        stub.chmod(stub.stat().st_mode | stat.S_IEXEC)

        def run(argv, with_stub):
            env = dict(os.environ)
            env["SPEAKERS_EMBEDDINGS_DIR"] = str(td / ("store_stub" if with_stub else "store"))
            env["PYTHONDONTWRITEBYTECODE"] = "1"
            # strip any real speaker_detection from PATH
            env["PATH"] = (str(stub_dir) + ":" if with_stub else "") + "/usr/bin:/bin"
            r = subprocess.run([sys.executable, str(REF / "speaker-assign")] + argv, capture_output=True, text=True, env=env)
            return r

        base = ["assign", str(wav), "-t", str(FIXTURE_SRC)]
        runs = [
            ("dry_json", ["--dry-run", "--format", "json"], False),
            ("expected", ["--dry-run", "--format", "json", "--expected-speakers", "alice,bob", "--threshold", "0.05"], False),
            ("emb_no_binary", ["--dry-run", "--format", "json", "--use-embeddings"], False),
            ("emb_stub", ["--dry-run", "--format", "json", "--use-embeddings"], True),
            ("emb_stub_medium", ["--dry-run", "--format", "json", "--use-embeddings", "--min-trust", "medium", "--threshold", "0.2"], True),
            ("emb_stub_high_ctx", ["--dry-run", "--format", "json", "--use-embeddings", "--min-trust", "high",
                                   "--expected-speakers", "bob,alice", "--context", "standup"], True),
            ("saved_json", ["--format", "json", "--use-embeddings", "--expected-speakers", "alice"], True),
        ]
        for name, extra, with_stub in runs:
            r = run(base + extra, with_stub)
            txt = r.stdout
            j = json.loads(txt[txt.index("{"):])
            j["transcript_path"] = "<TRANSCRIPT>"
            j.pop("assigned_at")
            rec = {"name": name, "argv": extra, "stub_rows": rows if with_stub else None, "rc": r.returncode, "json": j}
            if name == "saved_json":
                import yaml
                saved = td / "store_stub" / "assignments" / (j["recording_b3sum"] + ".yaml")
                y = yaml.safe_load(saved.read_text())
                y["transcript_path"] = "<TRANSCRIPT>"
                y.pop("assigned_at")
                rec["saved_yaml_obj"] = y
            cli.append(rec)
        import hashlib
        out["cli"] = {"wav_seed": 7, "wav_sha256_32": hashlib.sha256(wav.read_bytes()).hexdigest()[:32], "runs": cli}

    make_schema_golden(REF, HERE)
    (HERE / "plumbing_golden.json").write_text(json.dumps(out, indent=1, sort_keys=False) + "\n")
    print("wrote", HERE / "plumbing_golden.json", (HERE / "plumbing_golden.json").stat().st_size, "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
