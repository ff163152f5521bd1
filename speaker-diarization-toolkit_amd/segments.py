"""Transcript -> per-speaker time segments (mirror of speaker_detection_backends/transcript.py:25-305
and speaker_segments:38-71), plus the per-sentence splitter the GPU path uses so that
`identify` can finally score individual segments (speaker-assign:276-278 leaves `segments` unused).

Behaviour is pinned by golden vectors captured from the reference (tests/golden/plumbing_golden.json).
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, Iterator, List, Optional, Tuple

Segment = Tuple[float, float]
UNKNOWN_SPEAKER = "UU"


def load_transcript(path: Path) -> Dict[str, Any]:
    with open(path, "r") as fh:
        return json.load(fh)


def detect_transcript_format(data: Dict[str, Any]) -> str:
    """'assemblyai' | 'speechmatics' | 'unknown'  (transcript.py:25-53)."""
    if "utterances" in data:
        return "assemblyai"
    results = data.get("results")
    if isinstance(results, list) and results:
        head = results[0]
        if "alternatives" in head or "start_time" in head or head.get("type") in ("word", "punctuation"):
            return "speechmatics"
    return "unknown"


def _sm_words(data) -> Iterator[Tuple[str, float, float, str]]:
    """Speechmatics word items as (speaker, start, end, text); speaker falls back to the first
    alternative's, then to 'UU' (transcript.py:157-166)."""
    for item in data.get("results", []):
        if item.get("type") != "word":
            continue
        alts = item.get("alternatives", [])
        spk = item.get("speaker") or (alts[0].get("speaker") if alts else None) or UNKNOWN_SPEAKER
        text = alts[0].get("content", "") if alts else ""
        yield spk, item.get("start_time", 0), item.get("end_time", 0), text


def get_available_speakers(data: Dict[str, Any]) -> List[str]:
    fmt = detect_transcript_format(data)
    found = set()
    if fmt == "assemblyai":
        found.update(u["speaker"] for u in data.get("utterances", []) if "speaker" in u)
    elif fmt == "speechmatics":
        for item in data.get("results", []):
            if item.get("type") != "word":
                continue
            if "speaker" in item:
                found.add(item["speaker"])
            found.update(a["speaker"] for a in item.get("alternatives", []) if "speaker" in a)
    return sorted(found)


def _runs(data: Dict[str, Any], label: str) -> List[Dict[str, Any]]:
    """Contiguous runs of `label` with their text (transcript.py:191-263)."""
    fmt = detect_transcript_format(data)
    runs: List[Dict[str, Any]] = []
    if fmt == "assemblyai":
        for u in data.get("utterances", []):
            if u.get("speaker") == label:
                runs.append({"start": u.get("start", 0) / 1000.0, "end": u.get("end", 0) / 1000.0, "text": u.get("text", "")})
    elif fmt == "speechmatics":
        cur = None
        for spk, s, e, text in _sm_words(data):
            if spk == label:
                if cur is None:
                    cur = {"start": s, "end": e, "words": []}
                cur["end"] = e
                if text:
                    cur["words"].append(text)
            elif cur is not None:
                runs.append(cur)
                cur = None
        if cur is not None:
            runs.append(cur)
        runs = [{"start": r["start"], "end": r["end"], "text": " ".join(r["words"])} for r in runs]
    return runs


def extract_segments_as_tuples(data: Dict[str, Any], speaker_label: str) -> List[Segment]:
    """(start, end) per contiguous run, unmerged (transcript.py:123-188)."""
    return [(r["start"], r["end"]) for r in _runs(data, speaker_label)]


def extract_segments_from_transcript(data: Dict[str, Any], speaker_label: str, min_duration: float = 0.5,
                                     max_gap: float = 1.0) -> List[Dict[str, Any]]:
    """Runs shorter than min_duration dropped, then neighbours closer than max_gap merged
    (transcript.py:91-120, 266-286)."""
    out: List[Dict[str, Any]] = []
    for seg in _runs(data, speaker_label):
        if seg["end"] - seg["start"] < min_duration:
            continue
        if out and seg["start"] - out[-1]["end"] <= max_gap:
            out[-1]["end"] = seg["end"]
            if seg["text"]:
                out[-1]["text"] = (out[-1]["text"] + " " + seg["text"]).strip()
        else:
            out.append(dict(seg))
    return out


def merge_segments_by_gap(segments: List[Segment], max_gap: float) -> List[Segment]:
    """speaker_segments:38-71."""
    if not segments or max_gap <= 0:
        return segments
    out = [tuple(segments[0])]
    for s, e in segments[1:]:
        if s - out[-1][1] <= max_gap:
            out[-1] = (out[-1][0], e)
        else:
            out.append((s, e))
    return out


def sentence_segments(data: Dict[str, Any], speaker_label: Optional[str] = None) -> List[Dict[str, Any]]:
    """Sentence-level segments [{speaker, start, end}] split at Speechmatics `is_eos` punctuation
    (or per AssemblyAI utterance).  Not in the reference: this is the
    per-segment unit the GPU path embeds.  A sentence also ends when the speaker changes."""
    fmt = detect_transcript_format(data)
    out: List[Dict[str, Any]] = []
    if fmt == "assemblyai":
        for u in data.get("utterances", []):
            out.append({"speaker": u.get("speaker"), "start": u.get("start", 0) / 1000.0, "end": u.get("end", 0) / 1000.0})
    elif fmt == "speechmatics":
        cur = None
        for item in data.get("results", []):
            kind = item.get("type")
            if kind == "word":
                alts = item.get("alternatives", [])
                spk = item.get("speaker") or (alts[0].get("speaker") if alts else None) or UNKNOWN_SPEAKER
                if cur is not None and cur["speaker"] != spk:
                    out.append(cur)
                    cur = None
                if cur is None:
                    cur = {"speaker": spk, "start": item.get("start_time", 0), "end": item.get("end_time", 0)}
                else:
                    cur["end"] = item.get("end_time", cur["end"])
            elif kind == "punctuation" and item.get("is_eos") and cur is not None:
                out.append(cur)
                cur = None
        if cur is not None:
            out.append(cur)
    if speaker_label is not None:
        out = [s for s in out if s["speaker"] == speaker_label]
    return out
