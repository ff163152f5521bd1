#!/usr/bin/env python3
"""Where does the precise-mode fbank differ from the float64 oracle? (diagnostic)"""
import importlib, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from oracle import fbank as of
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
pcm = bench.synth_pcm(6, seed=3)
P = of.power_spectrum(pcm)
M = P @ of.mel_matrix()
Lo = 10.0 * np.log10(np.maximum(M, of.AMIN))                      # raw log-mel, float64
for prec in (1, 0):
    eng.set_precision(prec)
    feats = eng.fbank(torch.from_numpy(pcm).cuda())
    torch.cuda.synchronize()
    key = [k for k in eng._scratch if k.startswith("fbank")][0]
    L = eng._scratch[key][:6 * 201 * 80 * 4].view(torch.float32).reshape(6, 201, 80).cpu().numpy().astype(np.float64)
    err = np.abs(L - Lo)
    i = np.unravel_index(err.argmax(), err.shape)
    print(f"precision {prec}: raw log-mel |err| max {err.max():.3e} at seg/frame/mel {i} (value {Lo[i]:.3f} dB, mel power {M[i]:.3e}); median {np.median(err):.2e} p99 {np.percentile(err, 99):.2e} p99.9 {np.percentile(err, 99.9):.2e}")
    rel = np.abs(10 ** (L / 10) - M) / M
    print(f"   relative mel-power error: max {rel.max():.3e} median {np.median(rel):.2e}; by frame position: first frames {rel[:, :3].max():.2e} middle {rel[:, 50:150].max():.2e} last {rel[:, -3:].max():.2e}")
    print("   per-mel max rel err (every 10th):", " ".join(f"{rel[:, :, m].max():.1e}" for m in range(0, 80, 10)))
eng.set_precision(0)
