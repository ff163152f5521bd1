#!/usr/bin/env python3
"""Error budget of the PCM -> cosine-score path (VERDICT r2 next #1a): which rounding site of the bf16 layer-boundary model
 costs how much against the UN-ROUNDED model, on config #2's first N segments x 100 profiles.  CPU only:
everything here is the oracle (oracle/ecapa.py with per-site switches), nothing is the product.

    python tools/error_budget.py [--segments 64] [--out profiles/r03_error_budget]

Rows: every site ALONE at bf16, ALL-BUT-each, all (= the bf16 model the kernels implement), and the candidate precision modes:
every site at 11 / 16 / 22 significand bits (fp16, a bf16 hi+lo pair, an fp16 hi+lo pair), float32 instead of float64
accumulation, and the GPU fbank's split DFT table.  Columns: max |d score| over all pairs, over the top-1 scores, 1 - min cosine
between the embeddings, argmax IDs that differ.
"""
import argparse, importlib, json, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from oracle import ecapa as oe, fbank as of, scoring as osc


def run(weights, feats, P, sites, acc=torch.float64, chunk=16):
    m = oe.EcapaOracle(weights, "fp32", acc, sites=sites)
    out = [m.embed(feats[a:a + chunk]).numpy() for a in range(0, len(feats), chunk)]
    return oe.l2_normalise(np.concatenate(out))


def compare(E, Eref, P):
    S, Sref = osc.affinity(E, P).astype(np.float64), osc.affinity(Eref, P).astype(np.float64)
    i, iref = S.argmax(1), Sref.argmax(1)
    return {"max_abs_dscore_all_pairs": float(np.abs(S - Sref).max()),
            "max_abs_dscore_top1": float(np.abs(S.max(1) - Sref.max(1)).max()),
            "one_minus_min_cos": float(1.0 - (E.astype(np.float64) * Eref).sum(1).min()),
            "ids_differ": int((i != iref).sum())}


def budget(n_seg=64, n_prof=100, verbose=True, rows=None):
    weights = importlib.import_module("speaker-diarization-toolkit_amd.weights").synthetic_weights(0)
    pcm = bench.synth_pcm(n_seg, seed=0)
    P = bench.unit_rows(n_prof, 192, seed=1)
    feats = torch.from_numpy(of.fbank(pcm))
    S = oe.ROUNDING_SITES
    plan = [("none (reference: no rounding, float64 accumulation)", {}, torch.float64, feats)]
    plan += [(f"only {s} at bf16", {s: 8}, torch.float64, feats) for s in S]
    plan += [(f"all but {s} at bf16", {t: 8 for t in S if t != s}, torch.float64, feats) for s in S]
    plan += [("ALL sites at bf16 (= the model the kernels implement)", {t: 8 for t in S}, torch.float64, feats)]
    # which LAYERS' weights: every site at bf16, but one group of weights as a bf16 hi+lo pair (16 bits) - and the reverse
    allb = {t: 8 for t in S}
    plan += [(f"all at bf16, but the {g} weights as hi+lo pairs (16 bits)", dict(allb, **{f"w:{g}": 16}), torch.float64, feats) for g in oe.WEIGHT_GROUPS]
    plan += [("all at bf16, but the res2net + tdnn2 + tdnn1 weights as hi+lo pairs", dict(allb, **{"w:res2net": 16, "w:tdnn2": 16, "w:tdnn1": 16}), torch.float64, feats)]
    plan += [("all at bf16, ALL weights as hi+lo pairs EXCEPT res2net", dict(allb, w=16, **{"w:res2net": 8}), torch.float64, feats)]
    plan += [(f"ALL sites at {b} significand bits ({what})", {t: b for t in S}, torch.float64, feats)
             for b, what in ((11, "fp16 storage and operands"), (16, "bf16 hi+lo pairs"), (19, "for scale"), (22, "fp16 hi+lo pairs"))]
    plan += [("weights at 22 bits, activations exact (fp16 hi+lo weights, fp32 activations)", {"w": 22}, torch.float64, feats)]
    plan += [("no rounding, float32 accumulation (torch-CPU sgemm)", {}, torch.float32, feats)]
    plan += [("no rounding, fbank DFT table at 16 bits (the GPU fbank's hi+lo split)", {}, torch.float64, torch.from_numpy(of.fbank(pcm, dft_bits=16)))]
    if rows is not None:
        plan = [p for p in plan if p[0] in rows or p[0].startswith("none")]
    out, ref = [], None
    for name, sites, acc, f in plan:
        t0 = time.time()
        E = run(weights, f, P, sites, acc)
        if ref is None:
            ref = E
        r = dict(row=name, **compare(E, ref, P))
        out.append(r)
        if verbose:
            print(f"{name:85s} all {r['max_abs_dscore_all_pairs']:.3e}  top1 {r['max_abs_dscore_top1']:.3e}  1-cos {r['one_minus_min_cos']:.3e}  ids {r['ids_differ']}  ({time.time() - t0:.0f} s)", flush=True)
    return {"segments": n_seg, "profiles": n_prof, "workload": "config #2: synth_pcm(seed 0) first segments, unit_rows(100, seed 1), synthetic weights seed 0",
            "reference": "oracle/ecapa.py sites={} acc=float64 on oracle/fbank.py float64", "rows": out}


def to_markdown(rep):
    lines = [f"Error budget, {rep['segments']} config-#2 segments x {rep['profiles']} profiles, vs the un-rounded model (tools/error_budget.py)", "",
             "| rounding applied | max |d score| all pairs | top-1 | 1 - min cos(E) | IDs differ |", "|---|---|---|---|---|"]
    for r in rep["rows"][1:]:
        lines.append(f"| {r['row']} | {r['max_abs_dscore_all_pairs']:.2e} | {r['max_abs_dscore_top1']:.2e} | {r['one_minus_min_cos']:.2e} | {r['ids_differ']} |")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=64)
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r03_error_budget"))
    a = ap.parse_args()
    torch.set_num_threads(8)
    rep = budget(a.segments)
    Path(a.out + ".json").write_text(json.dumps(rep, indent=1))
    Path(a.out + ".md").write_text(to_markdown(rep))
    print("wrote", a.out + ".json/.md")
