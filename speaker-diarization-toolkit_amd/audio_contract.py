"""Input-audio contract (mirror of speaker_detection_backends/audio_profiles.py:12-111).

Same public names and behaviour - AudioProfile, PROFILES, get_profile, format_ffmpeg_args,
register_profile - checked against the reference's outputs in tests/test_plumbing_golden.py.
The GPU path consumes exactly the default profile: 16 kHz, mono, 16-bit PCM WAV.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

_WAV_CODECS = {8: "pcm_u8", 16: "pcm_s16le", 24: "pcm_s24le", 32: "pcm_s32le"}


@dataclass
class AudioProfile:
    sample_rate: int = 16000
    channels: int = 1
    format: str = "wav"
    bit_depth: int = 16
    max_duration_sec: Optional[float] = None


PROFILES: Dict[str, AudioProfile] = {
    "speechmatics": AudioProfile(),
    "pyannote": AudioProfile(),
    "mi355x": AudioProfile(),
    "default": AudioProfile(),
}


def get_profile(backend_name: str) -> AudioProfile:
    """Profile registered for a backend, else the default one (audio_profiles.py:50-60)."""
    return PROFILES.get(backend_name, PROFILES["default"])


def format_ffmpeg_args(profile: AudioProfile) -> List[str]:
    """ffmpeg conversion flags for a profile, without input/output paths (audio_profiles.py:63-100)."""
    out = ["-ar", str(profile.sample_rate), "-ac", str(profile.channels), "-f", profile.format]
    codec = _WAV_CODECS.get(profile.bit_depth) if profile.format == "wav" else None
    if codec:
        out += ["-acodec", codec]
    return out


def register_profile(name: str, profile: AudioProfile) -> None:
    PROFILES[name] = profile
