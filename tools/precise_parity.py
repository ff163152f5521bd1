#!/usr/bin/env python3
"""PCM -> score parity of BOTH numerical contracts on a larger sample than the test-suite's (default 256 config-#2 segments x 100 profiles),
against the un-rounded oracle with float64 accumulation, plus the near-tie sweep (profiles next to their perturbed twins) in both modes.
Checker side only (imports oracle/); prints one JSON object.   python tools/precise_parity.py [n_segments]"""
import importlib, json, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench
from oracle import ecapa as oecapa, fbank as ofbank
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
W = importlib.import_module("speaker-diarization-toolkit_amd.weights")
eng = ops.get_engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pcm = bench.synth_pcm(n, seed=0)
P = bench.unit_rows(100, 192, seed=1)
t0 = time.time()
model = oecapa.EcapaOracle(W.synthetic_weights(0), "fp32", torch.float64)
Eo = oecapa.l2_normalise(np.concatenate([model.embed(torch.from_numpy(ofbank.fbank(pcm[a:a + 32]))).numpy() for a in range(0, n, 32)]))
out = {"segments": n, "profiles": 100, "oracle": "oracle/ecapa.py fp32 mode, float64 accumulation", "oracle_seconds": round(time.time() - t0, 1)}


def top1(Et, Pm):
    Pn, Pb, rp = eng.l2norm(torch.from_numpy(np.ascontiguousarray(Pm, dtype=np.float32)).cuda())
    idx, sc = eng.affinity_topk(*Et, Pn, Pb, rp.max().reshape(1), k=1)
    torch.cuda.synchronize()
    return idx.cpu().numpy()[:, 0], sc.cpu().numpy()[:, 0]


for prec, name in ((0, "default"), (1, "precise")):
    eng.set_precision(prec)
    Et = eng.embed_pcm(torch.from_numpy(pcm).cuda())
    gi, gs = top1(Et, P)
    rep = bench.parity_object(Et[0].cpu().numpy(), gi, gs, Eo, P)
    rep.pop("mismatches", None)
    # near ties: every profile next to a perturbed twin, eps sweep; agreement of the argmax with the oracle's by fp32 margin
    rng = np.random.default_rng(7)
    edges = [0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, np.inf]
    tot = np.zeros(len(edges) - 1, np.int64); agree = np.zeros_like(tot); worst = 0.0
    for eps in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.1):
        q = rng.standard_normal(P.shape)
        twin = P.astype(np.float64) + eps * q / np.linalg.norm(q, axis=1, keepdims=True)
        Pj = np.concatenate([P, (twin / np.linalg.norm(twin, axis=1, keepdims=True)).astype(np.float32)]).astype(np.float32)
        So = Eo.astype(np.float64) @ Pj.astype(np.float64).T
        oi = So.argmax(1); srt = np.sort(So, axis=1); margin = srt[:, -1] - srt[:, -2]
        gj, _ = top1(Et, Pj)
        same = gj == oi
        if (~same).any(): worst = max(worst, float(margin[~same].max()))
        b = np.digitize(margin, edges) - 1
        for k in range(len(tot)):
            tot[k] += int((b == k).sum()); agree[k] += int((same & (b == k)).sum())
    rep["near_tie"] = {"margin_bins": [f"[{edges[k]:g}, {edges[k + 1]:g})" for k in range(len(tot))], "rows": tot.tolist(),
                       "agreement_rate": [round(float(a) / t, 4) if t else None for a, t in zip(agree, tot)], "largest_fp32_margin_of_a_changed_id": worst}
    out[name] = rep
eng.set_precision(0)
print(json.dumps(out))
