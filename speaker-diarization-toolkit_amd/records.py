"""Embedding records and profile checks for a local backend.

What the toolkit persists per enrollment is built inline in its CLI (speaker_detection:875-911) and checked by
speaker_detection_backends/schemas.py:140-251 (records) and :45-137 (profiles).  A backend that runs in-process - the
batch driver, the eval harness, the multi-GPU pipeline - needs the same record without going through that CLI, and must
never write one the toolkit's validator would complain about.  This module states both sides:

    make_embedding_record()   the dict cmd_enroll appends to profile["embeddings"][backend]   (speaker_detection:890-904)
    attach_embedding()        ... and the append itself                                         (speaker_detection:906-911)
    embedding_issues()        the findings validate_embedding() reports for a record           (schemas.py:140-251)
    profile_issues()          the findings validate_profile() reports for a profile            (schemas.py:45-137)

Findings are returned as the toolkit words them (tests/golden/schema_golden.json holds the toolkit's own output for the
same inputs); `strict=True` raises RecordError at the first finding the toolkit treats as fatal.
"""
from __future__ import annotations

import uuid
from datetime import datetime, timezone
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

TRUST_LEVELS = ("high", "invalidated", "low", "medium")          # sorted: the order the toolkit prints them in
SAMPLE_BUCKETS = ("reviewed", "unreviewed", "rejected")


class RecordError(ValueError):
    """A record or profile the toolkit's strict validation would reject."""


def trust_from_samples(samples: Dict[str, List[str]]) -> str:
    """speaker_detection:359-379: rejected -> invalidated; only reviewed -> high; some reviewed -> medium; else low."""
    if samples.get("rejected"):
        return "invalidated"
    if samples.get("reviewed"):
        return "medium" if samples.get("unreviewed") else "high"
    return "low"


def make_embedding_record(result: Dict[str, Any], audio_path: Path, audio_hash: str,
                          segments: Optional[Sequence[Tuple[float, float]]] = None,
                          samples: Optional[Dict[str, List[str]]] = None, trust_level: Optional[str] = None,
                          emb_id: Optional[str] = None, created_at: Optional[str] = None) -> Dict[str, Any]:
    """The record cmd_enroll stores for one `enroll_speaker` result (key order included).

    `result` is what the backend returned (only external_id, model_version and all_identifiers survive, as in the toolkit);
    `audio_hash` is the recording's b3sum / sha256[:32]; `samples` the sample hashes by review state (default: none tracked);
    `trust_level` overrides the level derived from `samples` (the CLI's --trust-level)."""
    samples = {k: list((samples or {}).get(k, [])) for k in SAMPLE_BUCKETS}
    rec: Dict[str, Any] = {
        "id": emb_id or f"emb-{uuid.uuid4().hex[:8]}",
        "external_id": result.get("external_id"),
        "source_audio": str(Path(audio_path).resolve()),
        "source_audio_b3sum": audio_hash,
        "source_segments": [{"start": s, "end": e} for s, e in segments] if segments else None,
        "model_version": result.get("model_version", "unknown"),
        "samples": samples,
        "trust_level": trust_level or trust_from_samples(samples),
        "created_at": created_at or datetime.now(timezone.utc).isoformat(),
    }
    if "all_identifiers" in result:
        rec["all_identifiers"] = result["all_identifiers"]
    return rec


def attach_embedding(profile: Dict[str, Any], backend_name: str, record: Dict[str, Any]) -> Dict[str, Any]:
    """Append a record to profile['embeddings'][backend] (created on demand); refuses a record the toolkit would reject."""
    embedding_issues(record, strict=True)
    profile.setdefault("embeddings", {}).setdefault(backend_name, []).append(record)
    return profile


def _kind(v: Any) -> str:
    return type(v).__name__


def embedding_issues(rec: Any, strict: bool = False) -> List[str]:
    found: List[str] = []

    def note(msg: str, fatal: bool) -> None:
        if fatal and strict:
            raise RecordError(msg)
        found.append(msg)

    if not isinstance(rec, dict):
        note(f"Embedding must be a dict, got {_kind(rec)}", True)
        return found
    absent = sorted({"id", "external_id", "created_at"} - rec.keys())
    if absent:
        note("Missing required fields: " + ", ".join(absent), True)
    if "id" in rec and not (isinstance(rec["id"], str) and rec["id"]):
        note("Embedding 'id' must be a non-empty string", True)
    ext = rec.get("external_id")
    if ext is not None and not isinstance(ext, str):
        note(f"Embedding 'external_id' must be a string or null, got {_kind(ext)}", True)
    if "model_version" in rec:
        mv = rec["model_version"]
        if not isinstance(mv, str):
            note(f"Embedding 'model_version' must be a string, got {_kind(mv)}", False)
        elif mv == "unknown":
            note("Embedding has unknown model_version", False)
    if "trust_level" in rec and rec["trust_level"] not in TRUST_LEVELS:
        note(f"Invalid trust_level '{rec['trust_level']}', expected one of: " + ", ".join(TRUST_LEVELS), True)
    if "created_at" in rec:
        stamp = rec["created_at"]
        if not isinstance(stamp, str):
            note(f"Embedding 'created_at' must be a string, got {_kind(stamp)}", False)
        else:
            try:
                datetime.fromisoformat(stamp.replace("Z", "+00:00"))
            except ValueError:
                note(f"Embedding 'created_at' is not valid ISO format: {stamp}", False)
    if "samples" in rec:
        sm = rec["samples"]
        if isinstance(sm, dict):
            for bucket in ("reviewed", "unreviewed", "rejected"):
                if bucket not in sm:
                    continue
                if not isinstance(sm[bucket], list):
                    note(f"samples.{bucket} must be a list", False)
                elif any(not isinstance(h, str) for h in sm[bucket]):
                    note(f"samples.{bucket} must contain only strings (b3sum hashes)", False)
        elif sm is not None:
            note(f"Embedding 'samples' must be a dict or null, got {_kind(sm)}", False)
    if "source_segments" in rec:
        segs = rec["source_segments"]
        if segs is not None and not isinstance(segs, list):
            note("Embedding 'source_segments' must be a list or null", False)
        for i, seg in enumerate(segs if isinstance(segs, list) else []):
            if not isinstance(seg, dict):
                note(f"source_segments[{i}] must be a dict", False)
            elif "start" not in seg or "end" not in seg:
                note(f"source_segments[{i}] must have 'start' and 'end' keys", False)
    return found


def profile_issues(prof: Any, strict: bool = False) -> List[str]:
    found: List[str] = []

    def note(msg: str, fatal: bool) -> None:
        if fatal and strict:
            raise RecordError(msg)
        found.append(msg)

    if not isinstance(prof, dict):
        note(f"Profile must be a dict, got {_kind(prof)}", True)
        return found
    absent = sorted({"id", "names"} - prof.keys())
    if absent:
        note("Missing required fields: " + ", ".join(absent), True)
    if "id" in prof and not (isinstance(prof["id"], str) and prof["id"]):
        note("Profile 'id' must be a non-empty string", True)
    if "names" in prof:
        if not isinstance(prof["names"], dict):
            note(f"Profile 'names' must be a dict, got {_kind(prof['names'])}", True)
        elif "default" not in prof["names"]:
            note("Profile 'names' should have a 'default' entry", False)
    if "tags" in prof:
        if not isinstance(prof["tags"], list):
            note(f"Profile 'tags' must be a list, got {_kind(prof['tags'])}", True)
        elif any(not isinstance(t, str) for t in prof["tags"]):
            note("All tags must be strings", True)
    if "embeddings" in prof:
        per_backend = prof["embeddings"]
        if not isinstance(per_backend, dict):
            note(f"Profile 'embeddings' must be a dict, got {_kind(per_backend)}", True)
        else:
            for backend, recs in per_backend.items():
                if not isinstance(recs, list):
                    note(f"Embeddings for '{backend}' must be a list", True)
                    continue
                for i, rec in enumerate(recs):
                    found.extend(f"embeddings.{backend}[{i}]: {w}" for w in embedding_issues(rec))
    if "version" in prof and not isinstance(prof["version"], int):
        note(f"Profile 'version' must be an int, got {_kind(prof['version'])}", False)
    return found
