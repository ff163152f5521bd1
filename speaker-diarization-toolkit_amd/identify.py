"""In-process counterpart of `speaker_detection identify <audio> --format json`
(speaker_detection:1031-1133) for the assignment driver: load db/*.json, filter, call the backend,
attach trust levels - without a subprocess per transcript label (speaker-assign:283-294)."""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path
from typing import Any, Dict, List, Optional

from .assign import rows_with_trust


def db_dir() -> Path:
    return Path(os.environ.get("SPEAKERS_EMBEDDINGS_DIR", os.path.expanduser("~/.config/speakers_embeddings"))) / "db"


def listing_pack_path() -> Path:
    """Where the parsed db listing is cached: $SPEAKERS_EMBEDDINGS_DIR/cache/profiles-pack.json - OUTSIDE db/, which this build shares with the
    reference CLI: that lists profiles with pathlib `db_path.glob("*.json")` (speaker_detection:213), which matches dot-files too, so any *.json
    kept inside db/ would be loaded there as a speaker profile without an 'id'."""
    return db_dir().parent / "cache" / "profiles-pack.json"


def list_all_speakers() -> List[Dict[str, Any]]:
    """Every db/*.json profile, sorted by file name (speaker_detection:206-220 reads them one by one, per CLI call).  At BASELINE's profile counts
    (1 000 / 10 000 speakers) that is 10^3-10^4 small-file reads per process, so the parsed list is kept as ONE file (listing_pack_path(), outside
    db/), keyed by a digest of the directory listing (name, size, mtime_ns, ctime_ns and inode of every db/*.json: one scandir, no file is opened):
    any edit, addition, deletion or replacement changes the digest - also a same-size rewrite inside the mtime granularity of a coarse file system,
    or by a tool that restores mtime, because ctime cannot be set from user space - and the pack is rebuilt from the files.  SDK_PROFILE_PACK=0
    disables it.  Warnings for unreadable files are repeated on a pack hit (they are stored with it)."""
    import hashlib
    d = db_dir()
    if not d.exists():
        return []
    entries = []
    for e in os.scandir(d):
        if e.name.endswith(".json") and e.is_file():                     # dot-files included: the reference's glob lists them too
            st = e.stat()
            entries.append((e.name, st.st_size, st.st_mtime_ns, st.st_ctime_ns, st.st_ino))
    entries.sort()
    use_pack = os.environ.get("SDK_PROFILE_PACK", "1") != "0" and len(entries) >= 64
    digest = hashlib.sha256(repr(entries).encode()).hexdigest()
    pack = listing_pack_path()
    if use_pack:
        try:
            t = json.loads(pack.read_text())
            if t.get("digest") == digest and t.get("format") == 2:
                for w in t.get("warnings", []):
                    print(w, file=sys.stderr)
                return t["profiles"]
        except (OSError, ValueError, KeyError):
            pass
    out, warns = [], []
    for name, *_ in entries:
        p = d / name
        try:
            out.append(json.loads(p.read_text()))
        except (json.JSONDecodeError, OSError) as exc:
            warns.append(f"Warning: Failed to load {p}: {exc}")
            print(warns[-1], file=sys.stderr)
    if use_pack:
        try:                                     # atomic publish; concurrent CLI processes write identical content
            pack.parent.mkdir(parents=True, exist_ok=True)
            tmp = pack.parent / f".profiles-pack.{os.getpid()}.tmp"
            tmp.write_text(json.dumps({"format": 2, "digest": digest, "profiles": out, "warnings": warns}))
            os.replace(tmp, pack)
        except OSError:
            pass
    return out


def candidates_for(backend_name: str, tags: Optional[List[str]] = None) -> List[Dict[str, Any]]:
    speakers = list_all_speakers()
    if tags:
        want = set(tags)
        speakers = [s for s in speakers if want <= set(s.get("tags", []))]   # AND logic (speaker_detection:241-243)
    return [s for s in speakers if s.get("embeddings", {}).get(backend_name)]


def make_rows_fn(audio_path: Path, tags: Optional[List[str]] = None, per_label: bool = False, threshold: float = 0.354,
                 backend=None, transcript: Optional[Dict[str, Any]] = None):
    """Return rows_fn(label, segments) for assign.assign_recording.  Whole-recording mode (the
    reference's behaviour) runs the GPU path once and serves every label from that result.

    per_label: the unit collect_embedding_signals was meant to use (its `segments` argument is accepted and ignored,
    speaker-assign:262-278).  With the parsed `transcript`, every label's `is_eos` sentences (segments.sentence_segments)
    are cut into true-length windows, ALL labels' windows are embedded in one bucketed GPU pass per recording, and each
    label is scored on its own windows only; without it the label's contiguous runs passed to rows_fn are used."""
    from .backend import Backend, aggregate_matches
    from .store import load_profile_batch
    be = backend or Backend()
    # the three rc-1 exits of cmd_identify before the backend is called (speaker_detection:1033-1057); speaker-assign
    # treats them as "no signals" (speaker-assign:296), so they become an empty row list with the same message
    if not Path(audio_path).exists():
        print(f"Error: Audio file not found: {audio_path}", file=sys.stderr)
        return lambda label, segs: []
    speakers = list_all_speakers()
    if tags:
        speakers = [s for s in speakers if set(tags) <= set(s.get("tags", []))]
    if not speakers:
        print("No speakers to match against.", file=sys.stderr)
        return lambda label, segs: []
    cands = [s for s in speakers if s.get("embeddings", {}).get(be.name)]
    by_id = {c["id"]: c for c in cands}
    if not cands:
        print(f"No speakers with {be.name} embeddings.", file=sys.stderr)
        return lambda label, segs: []
    if not per_label:
        try:
            rows = rows_with_trust(be.identify_speaker(audio_path, cands, threshold), by_id, be.name)
        except ValueError as exc:             # cmd_identify's `except Exception` -> stderr + rc 1 -> "no signals" upstream (speaker-assign:296)
            print(f"Error during identification: {exc}", file=sys.stderr)
            return lambda label, segs: []
        return lambda label, segs: rows

    from .wav import decode_to_profile
    batch = load_profile_batch(cands, be.name, model_prefix=f"{be.name}-", model_version=be.model_version, settings=be.numerics())
    for why in batch.skipped:
        print(f"mi355x backend: skipped embedding {why}", file=sys.stderr)
    if batch.all_skipped_message():           # loud, once: the in-process counterpart of the CLI's rc 1 (Backend.identify_speaker raises)
        print(f"Error during identification: {batch.all_skipped_message()}", file=sys.stderr)
    if len(batch) == 0:
        return lambda label, segs: []
    samples = decode_to_profile(Path(audio_path), be.engine(), be.get_audio_profile())

    def score(ranges):
        """ranges -> (best profile row, score, range index, span) per window"""
        idx, sc, wins = be.score_ranges(samples, ranges, batch)             # either host path (torch engine / SDK_NO_TORCH=1)
        if not wins:
            return [], [], []
        return idx[:, 0], sc[:, 0], wins

    if transcript is not None:
        from .segments import sentence_segments
        sents = sentence_segments(transcript)
        idx, sc, wins = score([(s["start"], s["end"]) for s in sents])          # one pass for the whole recording
        windows_of: Dict[str, List[int]] = {}
        for w, (ri, _, _) in enumerate(wins):
            windows_of.setdefault(sents[ri]["speaker"], []).append(w)

        def rows_fn(label, segs):
            ws = windows_of.get(label, [])
            if not ws:
                return []
            spans = [(wins[w][1], wins[w][2]) for w in ws]
            return rows_with_trust(aggregate_matches(idx[ws], sc[ws], spans, batch, threshold), by_id, be.name)
        rows_fn.windows = [(sents[ri]["speaker"], a, b) for ri, a, b in wins]    # for tests / -v output
        return rows_fn

    def rows_fn(label, segs):
        idx, sc, wins = score([(s["start"], s["end"]) for s in segs])
        if not len(wins):
            return []
        return rows_with_trust(aggregate_matches(idx, sc, [(a, b) for _, a, b in wins], batch, threshold), by_id, be.name)
    return rows_fn
