#!/usr/bin/env python3
"""Bias correction diagnostics: per layer, the measured channel means of its input and the size of the correction."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
W = importlib.import_module("speaker-diarization-toolkit_amd.weights")
WP = importlib.import_module("speaker-diarization-toolkit_amd.weights_pack")
eng = ops.Engine(0, bias_correction=True)
eff = eng.effective_weights()
plain = W.synthetic_weights(0)
for n, c, _ in WP.calib_layout()[0]:
    d = eff[f"{n}.conv.b"] - plain[f"{n}.conv.b"]
    print(f"{n:18s} |corr| max {np.abs(d).max():.3e} rms {np.sqrt((d**2).mean()):.3e}   |bias| rms {np.sqrt((plain[f'{n}.conv.b']**2).mean()):.3e}")
