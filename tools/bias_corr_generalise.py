#!/usr/bin/env python3
"""Does the bias correction (calibrated once on built-in noise + tones at T = 201) help on OTHER inputs?  PCM -> score deviation from the
un-rounded oracle (float32 accumulation), plain vs corrected default mode, on: config-#2 noise+tones (another seed), harmonic "voices",
quiet white noise, loud clipped noise, 0.5-s windows (T = 51) and 5-s windows (T = 501).  Checker side only (imports oracle/)."""
import importlib, json, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from oracle import ecapa as oecapa, fbank as ofbank, scoring as oscoring
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
W = importlib.import_module("speaker-diarization-toolkit_amd.weights")
plain, corr = ops.Engine(0, bias_correction=False), ops.Engine(0, bias_correction=True)
model = oecapa.EcapaOracle(W.synthetic_weights(0), "fp32", torch.float32)
P = bench.unit_rows(100, 192, seed=1)
rng = np.random.default_rng(77)


def voices(n, S):
    t = np.arange(S) / 16000.0
    out = []
    for i in range(n):
        f0 = rng.uniform(80, 260)
        x = sum((0.5 / h ** rng.uniform(1.0, 1.6)) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) for h in range(1, 12))
        x = x * (0.6 + 0.4 * np.sin(2 * np.pi * rng.uniform(2, 5) * t)) + rng.normal(0, 0.02, t.shape)
        out.append(np.clip(np.round(x / np.abs(x).max() * 0.5 * 32767), -32768, 32767).astype(np.int16))
    return np.stack(out)


n = 64
sets = {
    "config #2 noise + tones, seed 9 (T = 201)": bench.synth_pcm(n, seed=9),
    "harmonic voices (T = 201)": voices(n, 32000),
    "quiet white noise, sigma 0.01 (T = 201)": np.clip(np.round(rng.normal(0, 0.01, (n, 32000)) * 32768), -32768, 32767).astype(np.int16),
    "loud clipped noise, sigma 0.6 (T = 201)": np.clip(np.round(rng.normal(0, 0.6, (n, 32000)) * 32768), -32768, 32767).astype(np.int16),
    "0.5-s windows of config #2 (T = 51)": bench.synth_pcm(n, seed=10)[:, :8000].copy(),
    "5-s windows: config #2 x 2.5 (T = 501)": np.concatenate([bench.synth_pcm(16, seed=11), bench.synth_pcm(16, seed=12), bench.synth_pcm(16, seed=13)[:, :16000]], axis=1),
}
out = {}
for name, pcm in sets.items():
    Eo = oecapa.l2_normalise(np.concatenate([model.embed(torch.from_numpy(ofbank.fbank(pcm[a:a + 16]))).numpy() for a in range(0, len(pcm), 16)]))
    So = oscoring.affinity(Eo, P).astype(np.float64)
    row = {}
    for tag, eng in (("plain", plain), ("corrected", corr)):
        E = eng.embed_pcm(torch.from_numpy(pcm).cuda())[0].cpu().numpy()
        S = E.astype(np.float64) @ P.astype(np.float64).T
        row[tag] = {"max_abs_dscore": float(np.abs(S - So).max()), "ids_differ": int((S.argmax(1) != So.argmax(1)).sum()),
                    "one_minus_min_cos": float(1 - ((E.astype(np.float64) * Eo).sum(1) / (np.linalg.norm(E, axis=1) * np.linalg.norm(Eo, axis=1))).min())}
    row["improvement"] = round(row["plain"]["max_abs_dscore"] / row["corrected"]["max_abs_dscore"], 2)
    out[name] = row
    print(f"{name:48s} plain {row['plain']['max_abs_dscore']:.2e}  corrected {row['corrected']['max_abs_dscore']:.2e}  x{row['improvement']}  ids differ {row['plain']['ids_differ']}/{row['corrected']['ids_differ']}", flush=True)
print(json.dumps(out))
