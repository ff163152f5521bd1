"""embeddings/ on-disk store + batch loader.

The reference only mkdir-s, counts and rmtree-s `embeddings/<speaker_id>/*.npy`
(speaker_detection:85-87,598-600,615-619; speaker-report:292-294) - nothing there reads or writes
a vector.  This module gives the directory its content:

    embeddings/by-hash/<sha24>.npy        one float32 [192] unit vector per enrollment, named by
                                          content hash (enroll_speaker learns neither the speaker
                                          id nor the emb-id: speaker_detection:869-875)
    embeddings/<speaker_id>/<emb-id>.npy  hard link made at identify time, once both ids are known
                                          (keeps speaker-report's per-speaker count meaningful)

and the batch loader turns a list of candidate profile dicts (the `db/<id>.json` objects
cmd_identify passes down, speaker_detection:1054-1071) into ONE [P, 192] matrix + side tables.
"""
from __future__ import annotations

import hashlib
import os
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np

EXTERNAL_PREFIX = "npy:"
EMBED_DIM = 192
_KEY_RE = re.compile(r"[0-9a-f]{24}")          # vector_key(): the only thing that may follow "npy:" (it becomes a file name)


def embeddings_root() -> Path:
    root = Path(os.environ.get("SPEAKERS_EMBEDDINGS_DIR", os.path.expanduser("~/.config/speakers_embeddings")))
    return root / "embeddings"


def vector_key(vec: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(vec, dtype="<f4").tobytes()).hexdigest()[:24]


def save_vector(vec: np.ndarray, root: Optional[Path] = None) -> str:
    """Write one embedding; returns its external_id ('npy:<key>')."""
    vec = np.ascontiguousarray(vec, dtype=np.float32).reshape(-1)
    if vec.shape[0] != EMBED_DIM:
        raise ValueError(f"embedding has {vec.shape[0]} dims, expected {EMBED_DIM}")
    d = (root or embeddings_root()) / "by-hash"
    d.mkdir(parents=True, exist_ok=True)
    key = vector_key(vec)
    path = d / f"{key}.npy"
    if not path.exists():
        tmp = d / f".{key}.{os.getpid()}.tmp.npy"   # atomic publish: up to 4 CLI processes run at once
        np.save(tmp, vec)
        os.replace(tmp, path)
    return EXTERNAL_PREFIX + key


def vector_path(external_id: str, root: Optional[Path] = None) -> Path:
    if not isinstance(external_id, str) or not external_id.startswith(EXTERNAL_PREFIX):
        raise ValueError(f"not an mi355x external_id: {external_id!r}")
    key = external_id[len(EXTERNAL_PREFIX):]
    if not _KEY_RE.fullmatch(key):              # db/*.json is user-editable: never let the key walk out of by-hash/
        raise ValueError(f"malformed mi355x external_id: {external_id!r}")
    return (root or embeddings_root()) / "by-hash" / f"{key}.npy"


def load_vector(external_id: str, root: Optional[Path] = None) -> np.ndarray:
    v = np.load(vector_path(external_id, root), allow_pickle=False)
    if v.dtype != np.float32 or v.shape != (EMBED_DIM,):
        raise ValueError(f"{external_id}: stored array is {v.dtype}{v.shape}, expected float32[{EMBED_DIM}]")
    return v


def adopt(external_id: str, speaker_id: str, emb_id: str, root: Optional[Path] = None) -> None:
    """Link the by-hash file into embeddings/<speaker_id>/<emb-id>.npy (idempotent, best effort)."""
    src = vector_path(external_id, root)
    dst_dir = (root or embeddings_root()) / speaker_id
    dst = dst_dir / f"{emb_id}.npy"
    if dst.exists() or not src.exists():
        return
    dst_dir.mkdir(parents=True, exist_ok=True)
    try:
        os.link(src, dst)
    except OSError:
        import shutil
        shutil.copyfile(src, dst)


@dataclass
class ProfileBatch:
    """One row per enrolled embedding."""
    matrix: np.ndarray                       # [P, 192] float32 (unit rows as stored)
    speaker_ids: List[str] = field(default_factory=list)
    embedding_ids: List[Optional[str]] = field(default_factory=list)
    trust_levels: List[str] = field(default_factory=list)
    skipped: List[str] = field(default_factory=list)   # human-readable reasons

    def __len__(self) -> int:
        return len(self.speaker_ids)

    def all_skipped_message(self) -> Optional[str]:
        """One summary line when candidates were offered and NOT ONE of their vectors is usable (typically: every vector was
        enrolled under other weights) - the caller must make that loud instead of reporting "no match"."""
        if len(self) or not self.skipped:
            return None
        stale = sum("re-enroll" in why for why in self.skipped)
        return (f"{len(self.skipped)} of {len(self.skipped)} enrolled embeddings are unusable"
                + (f" ({stale} enrolled under other weights: re-enroll those speakers with the current model)" if stale else "")
                + f"; first: {self.skipped[0]}")


def load_profile_batch(candidates: List[Dict[str, Any]], backend_name: str, model_prefix: Optional[str] = None,
                       root: Optional[Path] = None, link: bool = True, model_version: Optional[str] = None) -> ProfileBatch:
    """Gather every usable embedding of every candidate into one matrix.
    Records with a foreign model_version prefix, a foreign external_id or a missing file are
    skipped with a reason (the caller logs them to stderr - never silently).
    `model_version` (exact) is stricter than the toolkit's prefix rule (base.py:92-93), which was written for a versioned
    remote API: a LOCAL model's version carries its weights digest, and a vector enrolled under other weights lives in a
    different embedding space - its cosines against the current model's embeddings are noise, so it is skipped too."""
    rows, sids, eids, trusts, skipped = [], [], [], [], []
    for prof in candidates:
        sid = prof.get("id")
        for rec in prof.get("embeddings", {}).get(backend_name, []) or []:
            ext, mv = rec.get("external_id"), rec.get("model_version", "unknown")
            tag = f"{sid}/{rec.get('id')}"
            if model_prefix and not str(mv).startswith(model_prefix):
                skipped.append(f"{tag}: model_version {mv} is not {model_prefix}*")
                continue
            if model_version and mv != model_version:
                skipped.append(f"{tag}: enrolled under {mv}, the loaded weights are {model_version} (re-enroll)")
                continue
            try:
                vec = load_vector(ext, root)
            except (ValueError, OSError) as exc:
                skipped.append(f"{tag}: {exc}")
                continue
            if link and sid and rec.get("id"):
                adopt(ext, sid, rec["id"], root)
            rows.append(vec)
            sids.append(sid)
            eids.append(rec.get("id"))
            trusts.append(rec.get("trust_level", "unknown"))
    mat = np.stack(rows).astype(np.float32) if rows else np.zeros((0, EMBED_DIM), np.float32)
    return ProfileBatch(mat, sids, eids, trusts, skipped)


def save_matrix_pack(path: Path, batch: ProfileBatch) -> None:
    """Optional packed form for large enrolments (10k+ profiles): one .npy + a JSON side table."""
    import json
    np.save(path, batch.matrix)
    Path(str(path) + ".json").write_text(json.dumps({"speaker_ids": batch.speaker_ids, "embedding_ids": batch.embedding_ids,
                                                     "trust_levels": batch.trust_levels}))


def load_matrix_pack(path: Path, mmap: bool = True) -> ProfileBatch:
    import json
    mat = np.load(path, mmap_mode="r" if mmap else None, allow_pickle=False)
    side = json.loads(Path(str(path) + ".json").read_text())
    return ProfileBatch(mat, side["speaker_ids"], side["embedding_ids"], side["trust_levels"])
