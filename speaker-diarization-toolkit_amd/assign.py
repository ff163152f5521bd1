"""speaker-assign compatible scoring / assignment driver (mirror of the reference CLI
`speaker-assign`: constants :49-70, parsers :169-246, collect_embedding_signals :262-328,
collect_context_signals :331-353, combine_signals :418-492, cmd_assign :499-649; and of
`speaker_detection identify`'s row post-processing, speaker_detection:1085-1127).

Differences that matter (all outside the arithmetic the goldens pin):
  * the embedding rows come from an in-process backend call (one GPU pass over all segments of
    the recording) instead of one `speaker_detection identify` subprocess per label;
  * with per_label=True each transcript label is scored on ITS OWN segments - the reference passes
    `segments` down and then ignores it (speaker-assign:276-278).

Float behaviour is deliberately identical: float64, accumulation in signal order, stable sort,
round(score, 3) - see tests/test_plumbing_golden.py.
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
from collections import defaultdict
from dataclasses import dataclass, field
from datetime import datetime, timezone
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional

VERSION = "1.0.0"
SCHEMA_VERSION = 1

SIGNAL_WEIGHTS = {"embedding_match": 0.4, "llm_name_detection": 0.3, "context_expected": 0.2,
                  "cross_backend_agreement": 0.1}
TRUST_MULTIPLIERS = {"high": 1.0, "medium": 0.7, "low": 0.4, "invalidated": 0.0, "unknown": 0.5}
CONFIDENCE_THRESHOLDS = {"high": 0.7, "medium": 0.4, "low": 0.2}
_TRUST_ORDER = ["low", "medium", "high"]
_TRUST_RANK = {"high": 3, "medium": 2, "low": 1, "unknown": 0, "invalidated": -1}


# ------------------------------------------------------------------------------- storage helpers
def get_speakers_embeddings_dir() -> Path:
    return Path(os.environ.get("SPEAKERS_EMBEDDINGS_DIR", os.path.expanduser("~/.config/speakers_embeddings")))


def compute_b3sum(file_path: Path) -> str:
    """First 32 hex chars of the file's BLAKE3 (via the `b3sum` tool) or, without it, SHA-256
    (speaker-assign:102-118)."""
    try:
        r = subprocess.run(["b3sum", "--no-names", str(file_path)], capture_output=True, text=True, check=True)
        return r.stdout.strip()[:32]
    except (subprocess.CalledProcessError, FileNotFoundError):
        h = hashlib.sha256()
        with open(file_path, "rb") as fh:
            for block in iter(lambda: fh.read(1 << 16), b""):
                h.update(block)
        return h.hexdigest()[:32]


def utc_now_iso() -> str:
    return datetime.now(timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ")


# ------------------------------------------------------------------------------- transcript side
def detect_transcript_format(data: dict) -> str:
    """speaker-assign:169-175 (looser than the backend parser: any 'results' key counts)."""
    if "utterances" in data:
        return "assemblyai"
    if "results" in data:
        return "speechmatics"
    return "unknown"


def get_speakers_from_transcript(data: dict) -> List[str]:
    fmt = detect_transcript_format(data)
    found = set()
    if fmt == "assemblyai":
        found.update(u["speaker"] for u in data.get("utterances", []) if u.get("speaker"))
    elif fmt == "speechmatics":
        for r in data.get("results", []):
            found.update(a["speaker"] for a in r.get("alternatives", []) if a.get("speaker"))
            if r.get("speaker"):
                found.add(r["speaker"])
    return sorted(found)


def get_speaker_segments(data: dict, speaker_label: str) -> List[dict]:
    """speaker-assign:199-246: consecutive results of one label form a segment; punctuation items
    carry the speaker too, so they extend segments (unlike the backend-side parser)."""
    fmt = detect_transcript_format(data)
    out: List[dict] = []
    if fmt == "assemblyai":
        for u in data.get("utterances", []):
            if u.get("speaker") == speaker_label:
                out.append({"start": u.get("start", 0) / 1000.0, "end": u.get("end", 0) / 1000.0, "text": u.get("text", "")})
    elif fmt == "speechmatics":
        cur = None
        for r in data.get("results", []):
            spk, text = None, ""
            for a in r.get("alternatives", []):
                if a.get("speaker"):
                    spk = a["speaker"]
                if a.get("content"):
                    text = a["content"]
            if r.get("speaker"):
                spk = r["speaker"]
            if spk == speaker_label and r.get("start_time") is not None:
                end = r.get("end_time", r["start_time"])
                if cur is None:
                    cur = {"start": r["start_time"], "end": end, "text": text}
                else:
                    cur["end"] = end
                    if text:
                        cur["text"] += " " + text
            elif cur is not None:
                out.append(cur)
                cur = None
        if cur is not None:
            out.append(cur)
    return out


# ------------------------------------------------------------------------------- signals
@dataclass
class Signal:
    type: str
    speaker_id: Optional[str]
    score: float
    evidence: dict = field(default_factory=dict)


@dataclass
class Assignment:
    speaker_label: str
    speaker_id: Optional[str]
    confidence: str
    score: float
    signals: List[dict]
    candidates: List[dict] = field(default_factory=list)


def signals_from_identify_rows(rows: Any, min_trust: str = "low") -> List[Signal]:
    """identify JSON rows -> embedding_match signals (speaker-assign:296-322)."""
    out: List[Signal] = []
    if not isinstance(rows, list):
        return out
    for row in rows:
        if not row.get("speaker_id"):
            continue
        trust = row.get("trust_level", "unknown")
        if min_trust in _TRUST_ORDER and trust in _TRUST_ORDER and _TRUST_ORDER.index(trust) < _TRUST_ORDER.index(min_trust):
            continue
        out.append(Signal("embedding_match", row["speaker_id"], row.get("score", 0.5),
                          {"embedding_id": row.get("embedding_id"), "trust_level": trust, "backend": row.get("backend")}))
    return out


def collect_context_signals(speaker_label: str, context_name: Optional[str], expected_speakers: List[str]) -> List[Signal]:
    return [Signal("context_expected", sid, 0.5, {"context": context_name, "reason": "in expected_speakers list"})
            for sid in expected_speakers]


def combine_signals(speaker_label: str, signals: List[Signal], threshold: float = 0.5) -> Assignment:
    """Weighted vote (speaker-assign:418-492).  Order-dependent float64 accumulation and a stable
    descending sort: the first-inserted speaker wins ties."""
    totals: Dict[str, float] = defaultdict(float)
    evidence: Dict[str, list] = defaultdict(list)
    for s in signals:
        if s.speaker_id is None:
            continue
        w = SIGNAL_WEIGHTS.get(s.type, 0.1)
        if s.type == "embedding_match":
            w *= TRUST_MULTIPLIERS.get(s.evidence.get("trust_level", "unknown"), 0.5)
        totals[s.speaker_id] += w * s.score
        evidence[s.speaker_id].append({"type": s.type, "score": s.score, **s.evidence})
    if not totals:
        return Assignment(speaker_label, None, "unassigned", 0.0, [], [])
    ranked = sorted(totals.items(), key=lambda kv: kv[1], reverse=True)
    best_id, best = ranked[0]
    band = "unassigned"
    for name in ("high", "medium", "low"):
        if best >= CONFIDENCE_THRESHOLDS[name]:
            band = name
            break
    as_dicts = lambda pairs: [{"speaker_id": sid, "score": sc} for sid, sc in pairs]  # noqa: E731
    if best < threshold:
        return Assignment(speaker_label, None, "unassigned", best, evidence.get(best_id, []), as_dicts(ranked[:3]))
    return Assignment(speaker_label, best_id, band, best, evidence.get(best_id, []), as_dicts(ranked[1:4]))


# ------------------------------------------------------------------------------- identify rows (a7)
def rows_with_trust(results: List[Dict[str, Any]], profiles_by_id: Dict[str, Dict[str, Any]], backend_name: str) -> List[Dict[str, Any]]:
    """Backend results -> the JSON rows `speaker_detection identify --format json` prints
    (speaker_detection:1085-1127): name, score == confidence, trust from the matched embedding or,
    without an embedding_id, the best trust among the speaker's embeddings."""
    rows = []
    for r in results:
        sid = r["speaker_id"]
        prof = profiles_by_id.get(sid)
        name = prof["names"]["default"] if prof and prof.get("names") else sid
        conf = r.get("confidence", r.get("similarity", 0))
        emb_id, trust = r.get("embedding_id"), "unknown"
        if prof:
            embs = prof.get("embeddings", {}).get(backend_name, [])
            if emb_id:
                for e in embs:
                    if e.get("id") == emb_id:
                        trust = e.get("trust_level", "unknown")
                        break
            elif embs:
                best_t, best_e = "unknown", None
                for e in embs:
                    t = e.get("trust_level", "unknown")
                    if _TRUST_RANK.get(t, 0) > _TRUST_RANK.get(best_t, 0):
                        best_t, best_e = t, e.get("id")
                trust, emb_id = best_t, best_e
        rows.append({"speaker_id": sid, "name": name, "score": conf, "confidence": conf, "trust_level": trust,
                     "embedding_id": emb_id, "backend": backend_name})
    return rows


def compute_trust_level(samples: Dict[str, List[str]]) -> str:
    """speaker_detection:359-379."""
    if samples.get("rejected"):
        return "invalidated"
    reviewed, unreviewed = samples.get("reviewed", []), samples.get("unreviewed", [])
    if reviewed and not unreviewed:
        return "high"
    return "medium" if reviewed else "low"


# ------------------------------------------------------------------------------- assignment (a11)
RowsFn = Callable[[str, List[dict]], Any]


def assign_recording(audio_path: Path, transcript_path: Path, *, rows_fn: Optional[RowsFn] = None,
                     use_embeddings: bool = False, min_trust: str = "low", context: Optional[str] = None,
                     expected_speakers: Optional[List[str]] = None, threshold: float = 0.3,
                     catalog_entry: Optional[dict] = None, b3sum: Optional[str] = None) -> Dict[str, Any]:
    """The body of cmd_assign (speaker-assign:499-616) as a function returning the output object.
    `rows_fn(label, segments)` supplies identify rows for a label (ignored unless use_embeddings)."""
    with open(transcript_path, "r") as fh:
        data = json.load(fh)
    labels = get_speakers_from_transcript(data)
    if not labels:
        raise ValueError("No speakers found in transcript")
    b3 = b3sum or compute_b3sum(Path(audio_path))
    expected: List[str] = []
    if catalog_entry:
        context = context or catalog_entry.get("context", {}).get("name")
        expected = catalog_entry.get("context", {}).get("expected_speakers", [])
    if expected_speakers:
        expected = list(expected_speakers)
    mappings: Dict[str, Any] = {}
    for label in labels:
        segs = get_speaker_segments(data, label)
        sigs: List[Signal] = []
        if use_embeddings and rows_fn is not None:
            sigs.extend(signals_from_identify_rows(rows_fn(label, segs), min_trust))
        if expected:
            sigs.extend(collect_context_signals(label, context, expected))
        a = combine_signals(label, sigs, threshold=threshold)
        entry = {"speaker_id": a.speaker_id, "confidence": a.confidence, "score": round(a.score, 3), "signals": a.signals}
        if a.candidates:
            entry["candidates"] = a.candidates
        mappings[label] = entry
    return {"schema_version": SCHEMA_VERSION, "recording_b3sum": b3, "transcript_path": str(transcript_path),
            "assigned_at": utc_now_iso(), "method": f"speaker-assign-v{VERSION}", "context": context,
            "min_trust": min_trust, "threshold": threshold, "mappings": mappings}


def save_assignment(output: Dict[str, Any], path: Optional[Path] = None) -> Path:
    """assignments/<b3sum>.yaml, block style, insertion order (speaker-assign:136-143, 628-629)."""
    import yaml
    if path is None:
        d = get_speakers_embeddings_dir() / "assignments"
        d.mkdir(parents=True, exist_ok=True)
        path = d / f"{output['recording_b3sum']}.yaml"
    with open(path, "w") as fh:
        yaml.dump(output, fh, default_flow_style=False, sort_keys=False, allow_unicode=True)
    return path


def main(argv: Optional[List[str]] = None) -> int:
    """`assign` sub-command with the reference's flags (speaker-assign:735-781); embedding rows come
    from the in-process MI355X backend.  Errors go to stderr with rc 1 (never a silent 'unassigned')."""
    import argparse
    ap = argparse.ArgumentParser(prog="speaker-assign-mi355x", description="Multi-signal speaker name assignment (MI355X backend)")
    sub = ap.add_subparsers(dest="command")
    a = sub.add_parser("assign")
    a.add_argument("audio")
    a.add_argument("--transcript", "-t", required=True)
    a.add_argument("--use-embeddings", "-e", action="store_true")
    a.add_argument("--min-trust", default="low", choices=["high", "medium", "low"])
    a.add_argument("--context", "-c")
    a.add_argument("--expected-speakers")
    a.add_argument("--tags")
    a.add_argument("--threshold", type=float, default=0.3)
    a.add_argument("--output", "-o")
    a.add_argument("--format", "-f", choices=["text", "json"], default="text")
    a.add_argument("--dry-run", "-n", action="store_true")
    a.add_argument("--per-label", action="store_true",
                   help="score each label on its own sentences (is_eos segments, true-length windows) instead of the whole recording")
    args = ap.parse_args(argv)
    if args.command != "assign":
        ap.print_help()
        return 0
    audio, transcript = Path(args.audio).resolve(), Path(args.transcript).resolve()
    for p, what in ((audio, "Audio"), (transcript, "Transcript")):
        if not p.exists():
            print(f"Error: {what} file not found: {p}", file=sys.stderr)
            return 1
    rows_fn = None
    if args.use_embeddings:
        from .identify import make_rows_fn
        try:
            with open(transcript, "r") as fh:
                parsed = json.load(fh) if args.per_label else None
            rows_fn = make_rows_fn(audio, tags=args.tags.split(",") if args.tags else None, per_label=args.per_label,
                                   transcript=parsed)
        except Exception as exc:  # noqa: BLE001
            print(f"Error during identification: {exc}", file=sys.stderr)
            return 1
    try:
        out = assign_recording(audio, transcript, rows_fn=rows_fn, use_embeddings=args.use_embeddings, min_trust=args.min_trust,
                               context=args.context, threshold=args.threshold,
                               expected_speakers=args.expected_speakers.split(",") if args.expected_speakers else None)
    except Exception as exc:  # noqa: BLE001
        print(f"Error: {exc}", file=sys.stderr)
        return 1
    if not args.dry_run:
        save_assignment(out)
        if args.output:
            save_assignment(out, Path(args.output))
    if args.format == "json":
        print(json.dumps(out, indent=2, ensure_ascii=False))
    else:
        for label, m in out["mappings"].items():
            print(f"  {label} -> {m.get('speaker_id') or '(unassigned)'} ({m['confidence']}, score: {m['score']:.2f})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
