"""Backend plug-in contract (mirror of speaker_detection_backends/base.py:22-304).

When the toolkit itself is importable, `EmbeddingBackend` IS the toolkit's ABC, so
`isinstance(Backend(), speaker_detection_backends.base.EmbeddingBackend)` holds and the toolkit's
CLIs load this backend through their own registry.  When it is not (the GPU box, unit tests), the
equivalent ABC below is used: same properties, same defaults, same error behaviour, pinned by
golden vectors captured from the reference (tests/golden/plumbing_golden.json: abc_defaults).
"""
from __future__ import annotations

import importlib
import os
import sys
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

from .audio_contract import AudioProfile, get_profile
from .segments import extract_segments_as_tuples, load_transcript

try:  # the real toolkit, when installed next to us
    from speaker_detection_backends.base import EmbeddingBackend as _ToolkitBackend  # type: ignore
except Exception:  # noqa: BLE001 - absent or unimportable: fall back to the mirror
    _ToolkitBackend = None


class _MirrorBackend(ABC):
    """Same surface as base.py:22-200."""

    @property
    @abstractmethod
    def name(self) -> str: ...

    @property
    @abstractmethod
    def requires_api_key(self) -> bool: ...

    @property
    def embedding_dim(self) -> Optional[int]:
        return None

    @property
    def model_version(self) -> str:
        return f"{self.name}-unknown"

    @property
    def audio_profile(self) -> Union[str, AudioProfile]:
        return "default"

    def get_audio_profile(self) -> AudioProfile:
        p = self.audio_profile
        return get_profile(p) if isinstance(p, str) else p

    def check_embedding_compatibility(self, embedding: Dict[str, Any]) -> Dict[str, Any]:
        version = embedding.get("model_version", "unknown")
        ok = version.startswith(f"{self.name}-")
        return {
            "compatible": ok,
            "version": version,
            "current": self.model_version,
            "warning": None if ok else (f"Embedding created with {version} may not work with "
                                        f"backend {self.name}. Consider re-enrolling."),
        }

    @abstractmethod
    def enroll_speaker(self, audio_path: Path, segments: Optional[List[Tuple[float, float]]] = None) -> Dict[str, Any]: ...

    @abstractmethod
    def identify_speaker(self, audio_path: Path, candidates: List[Dict[str, Any]], threshold: float = 0.354) -> List[Dict[str, Any]]: ...

    def verify_speaker(self, audio_path: Path, speaker_profile: Dict[str, Any], threshold: float = 0.354) -> Dict[str, Any]:
        hits = self.identify_speaker(audio_path, [speaker_profile], threshold)
        if not hits:
            return {"match": False, "similarity": 0.0, "embedding_id": None}
        return {"match": True, "similarity": hits[0]["similarity"], "embedding_id": hits[0].get("embedding_id")}

    def extract_segments_from_transcript(self, transcript_path: Path, speaker_label: str) -> List[Tuple[float, float]]:
        return extract_segments_as_tuples(load_transcript(transcript_path), speaker_label)


EmbeddingBackend = _ToolkitBackend if _ToolkitBackend is not None else _MirrorBackend
USING_TOOLKIT_ABC = _ToolkitBackend is not None

# ---------------------------------------------------------------------- registry (base.py:203-304)
PACKAGE = __name__.rsplit(".", 1)[0]
_DEFAULT_BACKENDS = {"mi355x": f"{PACKAGE}.backend"}
_LOADED: Optional[Dict[str, str]] = None


def _load_backends_config() -> Dict[str, str]:
    """$SPEAKER_BACKENDS_CONFIG, else backends.yaml beside this file, else the built-in default.
    Entry forms `name: {module: dotted.path}` and `name: dotted.path` (base.py:252-257)."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    path = None
    env = os.environ.get("SPEAKER_BACKENDS_CONFIG")
    if env:
        path = Path(env)
        if not path.exists():
            print(f"Warning: SPEAKER_BACKENDS_CONFIG not found: {path}", file=sys.stderr)
            path = None
    if path is None:
        here = Path(__file__).parent / "backends.yaml"
        path = here if here.exists() else None
    if path is not None:
        try:
            import yaml
            with open(path) as fh:
                data = yaml.safe_load(fh) or {}
            table = {}
            for name, info in (data.get("backends") or {}).items():
                if isinstance(info, dict):
                    table[name] = info.get("module", "")
                elif isinstance(info, str):
                    table[name] = info
            _LOADED = table
            return table
        except ImportError:
            pass
        except Exception as exc:  # noqa: BLE001 - same leniency as the reference
            print(f"Warning: Failed to load backends config: {exc}", file=sys.stderr)
    _LOADED = dict(_DEFAULT_BACKENDS)
    return _LOADED


def get_backend(name: str):
    table = _load_backends_config()
    if name not in table:
        raise ValueError(f"Unknown backend: {name}. Available: {', '.join(table.keys())}")
    return importlib.import_module(table[name]).Backend()


def list_backends() -> List[str]:
    return list(_load_backends_config().keys())


def reload_backends_config() -> None:
    global _LOADED
    _LOADED = None
