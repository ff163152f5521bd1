#!/usr/bin/env python3
"""HBM roofline of sdk_resample_s16: one hour of audio per call, HIP-event timing inside the library."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
for rate, ch in ((48000, 2), (44100, 2), (48000, 1), (8000, 1), (11025, 1)):
    n = rate * 3600
    x = torch.randint(-20000, 20000, (n, ch), dtype=torch.int16, device="cuda")
    for _ in range(2): eng.resample_s16(x, rate)
    eng.profile_begin()
    for _ in range(5): y = eng.resample_s16(x, rate)
    p = eng.profile_end()["resample"]
    ms = p["ms"] / 5
    print(f"{rate} Hz x{ch} -> 16000: {ms:.3f} ms per hour of audio, {p['bytes'] / 5 / ms / 1e6:.0f} GB/s algorithmic, "
          f"{p['flops'] / 5 / ms / 1e9:.2f} T int-MAC*2/s, {3600 / (ms * 1e-3):.3g} x real time", flush=True)
