#!/usr/bin/env python3
"""Backend-agnostic enrol -> identify accuracy check through the plug-in API.

Same case files and pass rule as the toolkit's evals/speaker_detection/benchmark.py:72-185 - a case names its
speakers' enrollment recordings, one test recording and the set of speakers expected in it; every speaker is
enrolled with `Backend.enroll_speaker(path)`, the test recording goes through `Backend.identify_speaker(path,
profiles, threshold=0.354)`, and the case passes when the identified set EQUALS the expected set.  Cases
001/002 are the toolkit's own definitions (evals/samples/*.test.json); 003 adds an enrolled-but-silent speaker.

The toolkit renders its audio with espeak-ng + ffmpeg (absent here).  `--synthesize` renders stand-in recordings
instead: one deterministic synthetic voice per `voice` tag (pitch, spectral tilt, vibrato), fresh noise/phase
seeds per recording, written as 16 kHz mono s16 WAVE.  With the random-initialised weights this repository
ships, the result says that the pipeline separates those voices end to end - not how a trained model performs
on speech; load trained weights (SDK_ECAPA_WEIGHTS=<npz>) and real recordings for that.

    python evals/run_eval.py --synthesize            # all cases, backend mi355x
    python evals/run_eval.py -t 001 -v --keep-temp
    python evals/run_eval.py --dry-run               # READY / MISSING per case, no GPU needed
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import shutil
import sys
import tempfile
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
sys.path.insert(0, str(ROOT))
PKG = "speaker-diarization-toolkit_amd"
THRESHOLD = 0.354                       # the toolkit's identify default (base.py:130-151)


def load_cases(samples_dir: Path, only: Optional[str] = None) -> List[Dict[str, Any]]:
    prefixes = [p.strip() for p in only.split(",")] if only else None
    cases = []
    for path in sorted(Path(samples_dir).glob("*.test.json")):
        case = json.loads(path.read_text())
        if prefixes and not any(case["id"].startswith(p) for p in prefixes):
            continue
        case["_path"] = str(path)
        cases.append(case)
    return cases


def audio_files(case: Dict[str, Any], audio_root: Path) -> Dict[str, Path]:
    files = {sid: audio_root / info["enrollment_audio"] for sid, info in case.get("speakers", {}).items()}
    files["<test>"] = audio_root / case["test_audio"]
    return files


def missing_audio(case: Dict[str, Any], audio_root: Path) -> List[str]:
    return [str(p) for p in audio_files(case, audio_root).values() if not p.exists()]


# ---------------------------------------------------------------------------- stand-in audio
def _voice_params(tag: str) -> Dict[str, float]:
    h = hashlib.sha256(tag.encode()).digest()
    return {"f0": 85.0 + 150.0 * h[0] / 255.0, "tilt": 0.9 + 0.8 * h[1] / 255.0, "vib": 2.0 + 4.0 * h[2] / 255.0,
            "formant": 500.0 + 1500.0 * h[3] / 255.0}


def render_voice(tag: str, seconds: float, seed: int, rate: int = 16000) -> np.ndarray:
    """Harmonic stack with a voice-specific pitch, tilt, vibrato and one resonance; int16."""
    v = _voice_params(tag)
    rng = np.random.default_rng(seed)
    t = np.arange(int(rate * seconds)) / rate
    f0 = v["f0"] * (1.0 + 0.01 * np.sin(2 * np.pi * v["vib"] * t))
    phase = 2 * np.pi * np.cumsum(f0) / rate
    x = np.zeros_like(t)
    for h in range(1, 25):
        fh = v["f0"] * h
        if fh > 0.45 * rate:
            break
        gain = h ** -v["tilt"] * (1.0 + 2.0 * np.exp(-((fh - v["formant"]) / 300.0) ** 2))
        x += gain * np.sin(h * phase + rng.uniform(0, 2 * np.pi))
    x *= 0.6 + 0.4 * np.sin(2 * np.pi * 3.3 * t + rng.uniform(0, 6.28))          # syllable-rate envelope
    x += rng.normal(0.0, 0.01 * np.abs(x).max(), x.shape)
    return np.clip(np.round(x / np.abs(x).max() * 0.5 * 32767), -32768, 32767).astype(np.int16)


def synthesize_case(case: Dict[str, Any], audio_root: Path) -> None:
    wav = importlib.import_module(f"{PKG}.wav")
    seed0 = int.from_bytes(hashlib.sha256(case["id"].encode()).digest()[:4], "little")
    for i, (sid, info) in enumerate(case["speakers"].items()):
        path = audio_root / info["enrollment_audio"]
        path.parent.mkdir(parents=True, exist_ok=True)
        if not path.exists():
            wav.write_wav_s16(path, render_voice(info.get("voice", sid), 8.0, seed=1000 + i))
    test = audio_root / case["test_audio"]
    if not test.exists():
        test.parent.mkdir(parents=True, exist_ok=True)
        takes = [render_voice(case["speakers"][sid].get("voice", sid), 5.0, seed=seed0 + 7 * j)
                 for j, sid in enumerate(case["expected_speakers"])]
        wav.write_wav_s16(test, np.concatenate(takes))


# ---------------------------------------------------------------------------- one case
def records_hash(path: Path) -> str:
    """sha256[:32] of the file: what the toolkit's compute_b3sum falls back to without the b3sum tool (speaker_detection:253-269)."""
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 16), b""):
            h.update(chunk)
    return h.hexdigest()[:32]


def run_case(case: Dict[str, Any], backend_name: str, audio_root: Path, work: Path, verbose: bool = False) -> Dict[str, Any]:
    res = {"test_id": case["id"], "passed": False, "enrolled": [], "identified": [], "expected": list(case["expected_speakers"]),
           "error": None, "scores": {}}
    say = (lambda m: print(m, file=sys.stderr)) if verbose else (lambda m: None)
    os.environ["SPEAKERS_EMBEDDINGS_DIR"] = str(work)
    (work / "db").mkdir(parents=True, exist_ok=True)
    try:
        backend = importlib.import_module(f"{PKG}.plugin_api").get_backend(backend_name)
    except Exception as e:                                    # unknown backend / missing library: the CLI's rc-1 path
        res["error"] = f"Backend error: {e}"
        return res
    profiles = []
    for sid, info in case.get("speakers", {}).items():
        path = audio_root / info["enrollment_audio"]
        try:
            enr = backend.enroll_speaker(path)
        except Exception as e:
            res["error"] = f"Enrollment failed for {sid}: {e}"
            return res
        prof = {"id": sid, "version": 1, "names": {"default": sid.title()}, "nicknames": [], "description": f"Test speaker {sid}",
                "metadata": {}, "tags": ["test"], "embeddings": {}}
        records = importlib.import_module(f"{PKG}.records")    # the record cmd_enroll would store (speaker_detection:890-904)
        records.attach_embedding(prof, backend_name, records.make_embedding_record(
            enr, path, records_hash(path), emb_id=f"emb-{sid}"))
        (work / "db" / f"{sid}.json").write_text(json.dumps(prof, indent=2))
        profiles.append(prof)
        res["enrolled"].append(sid)
        say(f"    enrolled {sid}")
    try:
        rows = backend.identify_speaker(audio_root / case["test_audio"], profiles, threshold=THRESHOLD)
    except Exception as e:
        res["error"] = f"Identification failed: {e}"
        return res
    res["identified"] = [r["speaker_id"] for r in rows]
    res["scores"] = {r["speaker_id"]: round(float(r.get("similarity", r.get("confidence", 0.0))), 4) for r in rows}
    say(f"    identified {res['scores']}")
    want, got = set(res["expected"]), set(res["identified"])
    res["passed"] = want == got
    if not res["passed"]:
        parts = []
        if want - got:
            parts.append(f"Missing speakers: {sorted(want - got)}")
        if got - want:
            parts.append(f"Extra speakers: {sorted(got - want)}")
        res["error"] = " ".join(parts)
    return res


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description="enrol/identify accuracy check through the EmbeddingBackend plug-in API")
    ap.add_argument("--tests", "-t", help="case id prefixes, comma separated")
    ap.add_argument("--backend", "-b", default=os.environ.get("SPEAKER_DETECTION_BACKEND", "mi355x"))
    ap.add_argument("--samples", default=str(HERE / "samples"), help="directory holding *.test.json")
    ap.add_argument("--audio-root", help="directory the cases' audio paths are relative to (default: a temp dir with --synthesize, else evals/)")
    ap.add_argument("--synthesize", action="store_true", help="render stand-in recordings for audio files that do not exist")
    ap.add_argument("--dry-run", "-n", action="store_true")
    ap.add_argument("--verbose", "-v", action="store_true")
    ap.add_argument("--keep-temp", action="store_true")
    ap.add_argument("--json", action="store_true", help="print the per-case results as one JSON document on stdout")
    a = ap.parse_args(argv)

    cases = load_cases(Path(a.samples), a.tests)
    if not cases:
        print("No test cases found.", file=sys.stderr)
        return 1
    temp_audio = None
    if a.audio_root:
        audio_root = Path(a.audio_root)
    elif a.synthesize and not a.dry_run:
        audio_root = temp_audio = Path(tempfile.mkdtemp(prefix="spk_eval_audio_"))
    else:
        audio_root = HERE
    out = sys.stderr if a.json else sys.stdout
    print(f"Speaker detection accuracy check\nBackend: {a.backend}\nTests: {len(cases)}\n", file=out)
    if a.dry_run:
        for c in cases:
            miss = missing_audio(c, audio_root)
            print(f"  {c['id']}: " + ("READY" if not miss else ("SYNTHESIZE" if a.synthesize else f"MISSING: {miss}")), file=out)
        return 0
    if a.synthesize:
        for c in cases:
            synthesize_case(c, audio_root)
    miss = [m for c in cases for m in missing_audio(c, audio_root)]
    if miss:
        print("Missing audio files (render them, or pass --synthesize):", file=out)
        for m in miss:
            print(f"  {m}", file=out)
        return 2
    results = []
    try:
        for c in cases:
            print(f"Test: {c['id']} - {c.get('description', '')}", file=out)
            work = Path(tempfile.mkdtemp(prefix=f"spk_eval_{c['id']}_"))
            try:
                r = run_case(c, a.backend, audio_root, work, a.verbose)
            finally:
                if not a.keep_temp:
                    shutil.rmtree(work, ignore_errors=True)
            results.append(r)
            print("  PASS" if r["passed"] else f"  FAIL: {r['error'] or 'Unknown error'}", file=out)
    finally:
        if temp_audio is not None and not a.keep_temp:
            shutil.rmtree(temp_audio, ignore_errors=True)
    ok = sum(r["passed"] for r in results)
    print(f"\nResults: {ok} passed, {len(results) - ok} failed", file=out)
    if a.json:
        print(json.dumps({"backend": a.backend, "passed": ok, "failed": len(results) - ok, "results": results}))
    return 0 if ok == len(results) else 1


if __name__ == "__main__":
    sys.exit(main())
