#!/usr/bin/env python3
"""VERDICT r4 next #3, the CPU half: would a single-plane fp16 default (11 significand bits at every rounding site, ONE MFMA per product - the
bf16 default's cost) with the bias correction be closer to the un-rounded model than the shipped bf16 default?  Oracle only (nothing here is the
product): oracle/ecapa.py with every site at `bits`, the bias correction restated for `bits` (b + (W - round(W, bits)) . mean of the layer input,
means from the oracle's own run of the built-in calibration audio), PCM -> score deviation from the un-rounded model on config #2's first
segments x 100 profiles and on the input kinds of tools/bias_corr_generalise.py.   python tools/fp16_decision.py [--segments 32]"""
import argparse, importlib, json, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from oracle import ecapa as oe, fbank as of, scoring as osc
W = importlib.import_module("speaker-diarization-toolkit_amd.weights")
WP = importlib.import_module("speaker-diarization-toolkit_amd.weights_pack")


class Recording(oe.EcapaOracle):
    """the oracle, noting the per-channel mean of every conv's input (what sdk_ecapa_forward_calib measures on the device)"""
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.sums, self.counts = {}, {}

    def conv(self, x, name, dilation=1):
        key = name[:-5] if name.endswith(".conv") else name
        s = x.double().sum(dim=(0, 1)).numpy()
        self.sums[key] = self.sums.get(key, 0) + s
        self.counts[key] = self.counts.get(key, 0) + x.shape[0] * x.shape[1]
        return super().conv(x, name, dilation)

    def means(self):
        return {k: v / self.counts[k] for k, v in self.sums.items()}


def corrected(weights, bits, means):
    out = dict(weights)
    for name, mu in means.items():
        if name == "blk0":
            continue
        w = weights[f"{name}.conv.w"].astype(np.float64)
        dw = w - oe.round_significand(torch.from_numpy(weights[f"{name}.conv.w"]), bits).double().numpy()
        corr = np.tensordot(dw.sum(axis=2), np.asarray(mu, np.float64)[:dw.shape[1]], axes=([1], [0]))
        out[f"{name}.conv.b"] = (weights[f"{name}.conv.b"].astype(np.float64) + corr).astype(np.float32)
    return out


def embed(model, feats, chunk=8):
    return oe.l2_normalise(np.concatenate([model.embed(feats[a:a + chunk]).numpy() for a in range(0, len(feats), chunk)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=32)
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r05_fp16_decision.json"))
    args = ap.parse_args()
    weights = W.synthetic_weights(0)
    P = bench.unit_rows(100, 192, seed=1)
    cal = torch.from_numpy(of.fbank(WP.calibration_pcm()))
    rng = np.random.default_rng(77)

    def voices(n, S):
        t = np.arange(S) / 16000.0
        out = []
        for _ in range(n):
            f0 = rng.uniform(80, 260)
            x = sum((0.5 / h ** rng.uniform(1.0, 1.6)) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) for h in range(1, 12))
            x = x * (0.6 + 0.4 * np.sin(2 * np.pi * rng.uniform(2, 5) * t)) + rng.normal(0, 0.02, t.shape)
            out.append(np.clip(np.round(x / np.abs(x).max() * 0.5 * 32767), -32768, 32767).astype(np.int16))
        return np.stack(out)
    n = args.segments
    sets = {"config #2 noise + tones (T = 201)": bench.synth_pcm(n, seed=0),
            "harmonic voices (T = 201)": voices(max(8, n // 2), 32000),
            "loud clipped noise, sigma 0.6 (T = 201)": np.clip(np.round(rng.normal(0, 0.6, (max(8, n // 4), 32000)) * 32768), -32768, 32767).astype(np.int16),
            "0.5-s windows of config #2 (T = 51)": bench.synth_pcm(max(8, n // 2), seed=10)[:, :8000].copy()}
    report = {"segments": n, "modes": {}}
    variants = {}
    for bits, label in ((8, "bf16"), (11, "fp16")):
        t0 = time.time()
        sites = {s: bits for s in oe.ROUNDING_SITES}
        rec = Recording(weights, "fp32", torch.float32, sites=sites)
        for a in range(0, len(cal), 8):
            rec.embed(cal[a:a + 8])
        variants[f"{label} plain"] = (weights, sites)
        variants[f"{label} bias-corrected"] = (corrected(weights, bits, rec.means()), sites)
        print(f"calibrated {label} in {time.time() - t0:.0f} s", flush=True)
    for sname, pcm in sets.items():
        feats = torch.from_numpy(of.fbank(pcm))
        ref = embed(oe.EcapaOracle(weights, "fp32", torch.float32, sites={}), feats)
        Sref = osc.affinity(ref, P).astype(np.float64)
        for vname, (w, sites) in variants.items():
            t0 = time.time()
            E = embed(oe.EcapaOracle(w, "fp32", torch.float32, sites=sites), feats)
            S = osc.affinity(E, P).astype(np.float64)
            row = {"max_abs_dscore_all_pairs": float(np.abs(S - Sref).max()), "ids_differ": int((S.argmax(1) != Sref.argmax(1)).sum()),
                   "one_minus_min_cos": float(1 - (E.astype(np.float64) * ref).sum(1).min())}
            report["modes"].setdefault(vname, {})[sname] = row
            print(f"{sname:44s} {vname:22s} max |d score| {row['max_abs_dscore_all_pairs']:.2e}  1 - cos {row['one_minus_min_cos']:.2e}  ids {row['ids_differ']}  ({time.time() - t0:.0f} s)", flush=True)
    Path(args.out).write_text(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
