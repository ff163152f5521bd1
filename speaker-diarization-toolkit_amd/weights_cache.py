"""On-disk cache of the PACKED device weight blob + digest (cold start, SURVEY.md section 7 "process model").

The toolkit constructs its backend afresh in every CLI process (speaker_detection_backends/base.py:272-293) and runs up to four at
once (speaker-process:627-629).  Without a cache each process generates or parses 20.8 M weights, SHA-256s 83 MB for the
model_version, re-packs the blob in numpy and only then uploads 42 MB.  With it, the second process maps one .npy and uploads.

Key = what identifies the weights WITHOUT reading them: for a checkpoint file (path, size, mtime_ns, sha256 of its first MiB), for the
seeded synthetic weights (seed, config).  Entry = <key>.p<precision>.npy (the blob, uint8) + <key>.json (digest, descriptor fields per
precision).  Written atomically (tmp + rename): four processes may race.  Everything here is plain numpy / JSON: nothing is unpickled.
SDK_WEIGHTS_CACHE=0 disables it, SDK_CACHE_DIR moves it (default ~/.cache/sdk_mi355x).
"""
from __future__ import annotations

import hashlib
import json
import os
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

FORMAT = 5          # bump when weights_pack.py's layout changes: old entries are then ignored


def enabled() -> bool:
    return os.environ.get("SDK_WEIGHTS_CACHE", "1") != "0"


def cache_dir() -> Path:
    return Path(os.environ.get("SDK_CACHE_DIR", os.path.expanduser("~/.cache/sdk_mi355x")))


def key_for_file(path: str) -> str:
    st = os.stat(path)
    h = hashlib.sha256()
    h.update(f"{FORMAT}|{os.path.abspath(path)}|{st.st_size}|{st.st_mtime_ns}|".encode())
    with open(path, "rb") as f:
        h.update(f.read(1 << 20))
    return "f" + h.hexdigest()[:24]


def key_for_seed(seed: int, cfg) -> str:
    return "s" + hashlib.sha256(f"{FORMAT}|seed{seed}|{cfg!r}".encode()).hexdigest()[:24]


def _meta_path(key: str) -> Path:
    return cache_dir() / f"{key}.json"


def load_meta(key: str) -> Optional[dict]:
    if not enabled():
        return None
    try:
        m = json.loads(_meta_path(key).read_text())
        return m if m.get("format") == FORMAT else None
    except (OSError, ValueError):
        return None


def load_blob(key: str, precision) -> Optional[Tuple[np.ndarray, dict]]:
    """(memory-mapped blob, descriptor fields) or None.  `precision`: 0 / 1, or "0c" for the default mode's bias-corrected blob."""
    m = load_meta(key)
    if not m or str(precision) not in m.get("fields", {}):
        return None
    try:
        blob = np.load(cache_dir() / f"{key}.p{precision}.npy", mmap_mode="r", allow_pickle=False)
    except (OSError, ValueError):
        return None
    f = m["fields"][str(precision)]
    if blob.dtype != np.uint8 or blob.ndim != 1 or blob.size != f.get("_bytes"):
        return None
    # the descriptor's offsets are dereferenced on the device: never trust an entry whose table points outside the blob, and never a blob
    # whose bytes are not the ones the table was written for (one SHA-256 pass over ~50 MB: 30 ms, against 2 s of generate + digest + pack)
    off = f.get("off")
    if (not isinstance(off, list) or len(off) != 256 or f.get("precision") != int(str(precision).split("c")[0])      # ("0c", "0c-<tag>" = default mode, bias-corrected)
            or any(not isinstance(o, int) or o < -1 or (o >= 0 and (o % 256 or o >= blob.size)) for o in off)):
        return None
    if hashlib.sha256(blob).hexdigest() != f.get("_sha256"):
        return None
    return blob, {k: v for k, v in f.items() if not k.startswith("_")}


def store(key: str, digest: str, precision: int, blob: np.ndarray, fields: dict) -> None:
    if not enabled():
        return
    try:
        d = cache_dir()
        d.mkdir(parents=True, exist_ok=True)
        tmp = d / f".{key}.p{precision}.{os.getpid()}.tmp.npy"
        np.save(tmp, np.ascontiguousarray(blob, dtype=np.uint8))
        os.replace(tmp, d / f"{key}.p{precision}.npy")
        # the meta file is shared by the entries of every precision: read-modify-write under an advisory lock, so two processes storing
        # different entries (default / corrected / precise) at once cannot drop each other's fields
        import fcntl
        with open(d / f".{key}.lock", "a+") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            m = load_meta(key) or {"format": FORMAT, "digest": digest, "fields": {}}
            m["digest"] = digest
            m["fields"][str(precision)] = dict(fields, _bytes=int(blob.size), _sha256=hashlib.sha256(np.ascontiguousarray(blob, dtype=np.uint8)).hexdigest())
            tmpj = d / f".{key}.{os.getpid()}.tmp.json"
            tmpj.write_text(json.dumps(m))
            os.replace(tmpj, _meta_path(key))
    except OSError:
        pass            # a read-only home directory must not break the backend: the cache is an optimisation
