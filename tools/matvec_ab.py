#!/usr/bin/env python3
"""A/B of the config #5 tile kernel: matvec_variant 0 (round 3, persistent row groups) vs 1 (round 1), interleaved, HIP events."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
for N, rows in ((100_000, 100_000), (100_000, 12_500), (20_000, 20_000)):
    k = 16
    E, Eb, _ = eng.l2norm(torch.randn(N, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)))
    X = torch.randn(N, k, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
    res = {}
    for rep in range(3):
        for v in (0, 1):
            eng.set_option("matvec_variant", v)
            eng.affinity_matvec(Eb, X, 0, rows)
            eng.profile_begin()
            for _ in range(3): Y = eng.affinity_matvec(Eb, X, 0, rows)
            ms = eng.profile_end()["affinity_matvec"]["ms"] / 3
            res.setdefault(v, []).append(ms)
            if rep == 0: res[f"Y{v}"] = Y.clone()
    eng.set_option("matvec_variant", 0)
    d = float((res["Y0"] - res["Y1"]).abs().max() / res["Y1"].abs().max())
    fl = 2.0 * rows * N * (192 + k)
    print(f"N={N} rows={rows}: new {min(res[0]):.3f} ms ({fl / min(res[0]) / 1e9:.0f} TF, frac {fl / min(res[0]) / 1e9 / 2500:.3f})  old {min(res[1]):.3f} ms ({fl / min(res[1]) / 1e9:.0f} TF)  all: {[round(x, 3) for x in res[0]]} vs {[round(x, 3) for x in res[1]]}  max rel diff {d:.2e}")
