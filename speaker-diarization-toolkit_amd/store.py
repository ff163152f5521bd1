"""embeddings/ on-disk store + batch loader.

The reference only mkdir-s, counts and rmtree-s `embeddings/<speaker_id>/*.npy`
(speaker_detection:85-87,598-600,615-619; speaker-report:292-294) - nothing there reads or writes
a vector.  This module gives the directory its content:

    embeddings/by-hash/<sha24>.npy        one float32 [192] unit vector per enrollment, named by
                                          content hash (enroll_speaker learns neither the speaker
                                          id nor the emb-id: speaker_detection:869-875)
    embeddings/<speaker_id>/<emb-id>.npy  hard link made at identify time, once both ids are known
                                          (keeps speaker-report's per-speaker count meaningful)

    embeddings/packs/pack-<model>-<set>.npy  PACKED profile matrix of one candidate set (+ .json side table): the fp32 rows, their
                                          L2-normalised bf16 copy and the rounding residuals as sdk_l2norm wrote them, so a
                                          later process maps ONE file, uploads and scores - no per-embedding file I/O
                                          (the reference loads every db/*.json per call, speaker_detection:206-220,1054;
                                          at BASELINE's 1k / 10k profiles one np.load + one link probe per embedding is
                                          10^4 small-file opens per CLI process)

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


def save_vector(vec: np.ndarray, root: Optional[Path] = None, meta: Optional[Dict[str, Any]] = None) -> str:
    """Write one embedding; returns its external_id ('npy:<key>').  meta (optional): the numerical setting it was made under (Backend: bias
    correction on / off, precision), kept beside it as <key>.meta.json - model_version names the weights, not the setting, and vectors made under
    another setting are COMPARABLE but differ at the ~4e-3 level, so identify warns instead of skipping them."""
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
    if meta:
        mp = d / f"{key}.meta.json"
        if not mp.exists():
            import json
            tmpm = d / f".{key}.{os.getpid()}.tmp.json"
            tmpm.write_text(json.dumps(meta, sort_keys=True))
            os.replace(tmpm, mp)
    return EXTERNAL_PREFIX + key


def load_vector_meta(external_id: str, root: Optional[Path] = None) -> Optional[Dict[str, Any]]:
    import json
    try:
        return json.loads(vector_path(external_id, root).with_suffix("").with_suffix(".meta.json").read_text())
    except (OSError, ValueError):
        return None


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
    except FileExistsError:
        return                                   # another CLI process linked it between the probe and here (speaker-process runs four at once)
    except OSError:
        import shutil
        tmp = dst_dir / f".{emb_id}.{os.getpid()}.tmp"
        try:                                     # no hard links on this file system: copy, published atomically
            shutil.copyfile(src, tmp)
            os.replace(tmp, dst)
        except OSError:
            pass


_BATCH_UID = __import__("itertools").count(1)


@dataclass
class ProfileBatch:
    """One row per enrolled embedding."""
    matrix: np.ndarray                       # [P, 192] float32 (unit rows as stored)
    speaker_ids: List[str] = field(default_factory=list)
    embedding_ids: List[Optional[str]] = field(default_factory=list)
    trust_levels: List[str] = field(default_factory=list)
    skipped: List[str] = field(default_factory=list)   # human-readable reasons
    warnings: List[str] = field(default_factory=list)  # usable, but enrolled under another numerical setting (bias correction / precision)
    # packed store (see pack_*): `norm` = (E fp32 [P,192], Eb bf16 bits uint16 [P,192], resid fp32 [P]) exactly as sdk_l2norm produced them when
    # the pack was built (pack hit: memory-mapped, upload and score; None: normalise on the device); `pack_ref` = where to publish them after
    # the first normalisation of a batch that was loaded file by file (pack miss); `from_pack` says which of the two happened
    norm: Optional[tuple] = None
    pack_ref: Optional[tuple] = None
    from_pack: bool = False
    linked: bool = False               # the loader that built this batch made the per-speaker hard links (recorded in the pack's side table)
    uid: int = field(default_factory=lambda: next(_BATCH_UID))      # process-unique identity (device-copy reuse across scoring calls of one batch)

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


# ---------------------------------------------------------------------------------------------------------------- packed profile matrix (k7)
PACK_FORMAT = 2      # 2: the side table carries crc32 of the blob and whether the per-speaker links were made
PACK_KEEP = 8          # packs kept per store (one per candidate set x model; the oldest are pruned at publish time)


def pack_enabled() -> bool:
    return os.environ.get("SDK_PROFILE_PACK", "1") != "0"


def pack_min_rows() -> int:
    """Below this many embeddings the per-file loader is as fast as the pack (default 16)."""
    return max(1, int(os.environ.get("SDK_PROFILE_PACK_MIN", "16")))


def candidate_digest(candidates: List[Dict[str, Any]], backend_name: str, model_prefix: Optional[str], model_version: Optional[str],
                     settings: Optional[Dict[str, Any]] = None) -> str:
    """Identity of a candidate set as the loader sees it: every (speaker id, embedding id, external_id, model_version, trust level) of the
    backend's records, in order, plus the filter arguments.  The by-hash files are content-addressed (external_id = hash of the vector), so
    the keys alone pin the matrix: no stat() per embedding is needed to notice a change - a re-enrolment changes a key, a deletion removes one."""
    h = hashlib.sha256(f"pack{PACK_FORMAT}|{backend_name}|{model_prefix}|{model_version}|{EMBED_DIM}|{sorted((settings or {}).items())}\n".encode())
    for prof in candidates:
        sid = prof.get("id")
        for rec in prof.get("embeddings", {}).get(backend_name, []) or []:
            h.update(f"{sid}|{rec.get('id')}|{rec.get('external_id')}|{rec.get('model_version', 'unknown')}|{rec.get('trust_level', 'unknown')}\n".encode())
    return h.hexdigest()[:20]


def _pack_paths(root: Optional[Path], model_version: Optional[str], digest: str):
    tag = hashlib.sha256(str(model_version).encode()).hexdigest()[:10]
    d = (root or embeddings_root()) / "packs"
    return d / f"pack-{tag}-{digest}.npy", d / f"pack-{tag}-{digest}.json"


def load_pack(candidates: List[Dict[str, Any]], backend_name: str, model_prefix: Optional[str] = None, root: Optional[Path] = None,
              model_version: Optional[str] = None, settings: Optional[Dict[str, Any]] = None, link: bool = False) -> Optional[ProfileBatch]:
    """The packed matrix of exactly this candidate set, memory-mapped, or None (no pack, stale format, truncated or corrupted file: the side
    table carries the blob's crc32).  One JSON read + one mmap + one checksum sweep whatever P is.  link: make the embeddings/<speaker_id>/<emb-id>.npy
    hard links if the builder of the pack did not (it was called with link=False): once, recorded in the side table."""
    import json
    import zlib
    digest = candidate_digest(candidates, backend_name, model_prefix, model_version, settings)
    npy, side = _pack_paths(root, model_version, digest)
    try:
        t = json.loads(side.read_text())
        if t.get("format") != PACK_FORMAT or t.get("digest") != digest or t.get("dim") != EMBED_DIM:
            return None
        P = int(t["rows"])
        blob = np.load(npy, mmap_mode="r", allow_pickle=False)
    except (OSError, ValueError, KeyError, TypeError):
        return None
    per_row = 2 * EMBED_DIM * 4 + EMBED_DIM * 2 + 4          # layout: [matrix fp32 P x 192][E fp32 P x 192][Eb bf16 bits P x 192][resid fp32 P]
    if blob.dtype != np.uint8 or blob.ndim != 1 or blob.size != P * per_row or any(len(t.get(k, ())) != P for k in ("speaker_ids", "embedding_ids", "trust_levels")):
        return None
    if zlib.crc32(blob) != t.get("crc32"):
        return None
    if link and not t.get("linked"):
        # the pack was built by a caller that did not link (or could not): the cheap pass the per-file loader runs per record, once per pack
        ext_of = {(prof.get("id"), rec.get("id")): rec.get("external_id") for prof in candidates for rec in prof.get("embeddings", {}).get(backend_name, []) or []}
        for sid, eid in zip(t["speaker_ids"], t["embedding_ids"]):
            if sid and eid and ext_of.get((sid, eid)):
                adopt(ext_of[(sid, eid)], sid, eid, root)
        try:
            t["linked"] = True
            tmpj = side.parent / f".{side.stem}.{os.getpid()}.tmp.json"
            tmpj.write_text(json.dumps(t))
            os.replace(tmpj, side)
        except OSError:
            pass
    return _pack_batch(blob, P, t)


def _pack_batch(blob: np.ndarray, P: int, t: dict) -> ProfileBatch:
    a = P * EMBED_DIM * 4
    mat = blob[:a].view(np.float32).reshape(P, EMBED_DIM)
    E = blob[a:2 * a].view(np.float32).reshape(P, EMBED_DIM)
    Eb = blob[2 * a:2 * a + P * EMBED_DIM * 2].view(np.uint16).reshape(P, EMBED_DIM)
    r = blob[2 * a + P * EMBED_DIM * 2:].view(np.float32)
    return ProfileBatch(mat, list(t["speaker_ids"]), list(t["embedding_ids"]), list(t["trust_levels"]), list(t.get("skipped", [])),
                        warnings=list(t.get("warnings", [])), norm=(E, Eb, r), from_pack=True)


def publish_pack(batch: ProfileBatch, E: np.ndarray, Eb_bits: np.ndarray, resid: np.ndarray) -> Optional[Path]:
    """Write the pack of a batch that was loaded file by file (batch.pack_ref), with the normalised copies the device produced for it.
    Atomic (tmp + rename, the side table LAST so a reader never sees a table without its matrix); concurrent builders of the same set write
    identical bytes and the last rename wins (speaker-process:627-629 runs up to 4 CLI processes at once).  Best effort: a read-only store
    must not break identify."""
    import json
    import zlib
    if not batch.pack_ref:
        return None
    npy, side, digest = batch.pack_ref
    P = len(batch)
    E = np.ascontiguousarray(E, dtype=np.float32)
    Eb_bits = np.ascontiguousarray(Eb_bits, dtype=np.uint16)
    resid = np.ascontiguousarray(resid, dtype=np.float32)
    if E.shape != (P, EMBED_DIM) or Eb_bits.shape != (P, EMBED_DIM) or resid.shape != (P,):
        raise ValueError(f"publish_pack: normalised arrays {E.shape} {Eb_bits.shape} {resid.shape} do not match {P} profiles")
    try:
        npy.parent.mkdir(parents=True, exist_ok=True)
        blob = np.concatenate([np.ascontiguousarray(batch.matrix, dtype=np.float32).view(np.uint8).reshape(-1), E.view(np.uint8).reshape(-1),
                               Eb_bits.view(np.uint8).reshape(-1), resid.view(np.uint8).reshape(-1)])
        tmp = npy.parent / f".{npy.stem}.{os.getpid()}.tmp.npy"
        np.save(tmp, blob)
        os.replace(tmp, npy)
        tmpj = npy.parent / f".{side.stem}.{os.getpid()}.tmp.json"
        tmpj.write_text(json.dumps({"format": PACK_FORMAT, "digest": digest, "dim": EMBED_DIM, "rows": P, "speaker_ids": batch.speaker_ids,
                                    "embedding_ids": batch.embedding_ids, "trust_levels": batch.trust_levels, "skipped": batch.skipped,
                                    "warnings": batch.warnings, "crc32": zlib.crc32(blob), "linked": bool(batch.linked)}))
        os.replace(tmpj, side)
        def age(q):                                                        # another CLI process may prune the same file between glob and stat
            try:
                return q.stat().st_mtime
            except OSError:
                return 0.0
        packs = sorted(npy.parent.glob("pack-*.json"), key=age, reverse=True)
        for old in packs[PACK_KEEP:]:                                      # bounded: one pack per (candidate set, model)
            for q in (old, old.with_suffix(".npy")):
                try:
                    q.unlink()
                except OSError:
                    pass
    except OSError:
        return None
    batch.pack_ref = None
    return npy


def load_profile_batch(candidates: List[Dict[str, Any]], backend_name: str, model_prefix: Optional[str] = None,
                       root: Optional[Path] = None, link: bool = True, model_version: Optional[str] = None,
                       use_pack: Optional[bool] = None, settings: Optional[Dict[str, Any]] = None) -> ProfileBatch:
    """Gather every usable embedding of every candidate into one matrix.
    use_pack (default: $SDK_PROFILE_PACK != 0): serve the set from its packed matrix when one exists (`from_pack`, `norm` set: ONE file mapped,
    no per-embedding I/O); otherwise load file by file and leave `pack_ref` set so the caller publishes the pack after the first device
    normalisation (publish_pack).
    Records with a foreign model_version prefix, a foreign external_id or a missing file are
    skipped with a reason (the caller logs them to stderr - never silently).
    `model_version` (exact) is stricter than the toolkit's prefix rule (base.py:92-93), which was written for a versioned
    remote API: a LOCAL model's version carries its weights digest, and a vector enrolled under other weights lives in a
    different embedding space - its cosines against the current model's embeddings are noise, so it is skipped too."""
    use_pack = pack_enabled() if use_pack is None else use_pack
    if use_pack:
        hit = load_pack(candidates, backend_name, model_prefix, root, model_version, settings, link=link)
        if hit is not None:
            return hit
    rows, sids, eids, trusts, skipped, warns = [], [], [], [], [], []
    unreadable = 0
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
                unreadable += 1
                continue
            if link and sid and rec.get("id"):
                adopt(ext, sid, rec["id"], root)
            if settings:
                made = load_vector_meta(ext, root)
                if made and any(made.get(k) != v for k, v in settings.items() if k in made):
                    warns.append(f"{tag}: enrolled with {made}, this process runs {settings} - comparable, but scores differ at the ~4e-3 level (re-enroll for "
                                 "the tightest match)")
            rows.append(vec)
            sids.append(sid)
            eids.append(rec.get("id"))
            trusts.append(rec.get("trust_level", "unknown"))
    mat = np.stack(rows).astype(np.float32) if rows else np.zeros((0, EMBED_DIM), np.float32)
    batch = ProfileBatch(mat, sids, eids, trusts, skipped, warns)
    # a set with an unreadable / missing vector file is not packed: the digest is over the records' keys, so the file turning up later (a store
    # copied after its database) would not invalidate a pack that had recorded it as skipped
    if use_pack and len(batch) >= pack_min_rows() and not unreadable:
        digest = candidate_digest(candidates, backend_name, model_prefix, model_version, settings)
        batch.pack_ref = _pack_paths(root, model_version, digest) + (digest,)
        batch.linked = bool(link)
    return batch
