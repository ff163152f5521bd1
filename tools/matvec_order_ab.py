#!/usr/bin/env python3
"""Round 4: the mat-vec's range order (ranges sorted by start stage, 32 neighbours per XCD: matvec_variant 0) against the plain map blockIdx -> range
(matvec_variant 2), interleaved, HIP events; Y must be bit-identical (the work of a range does not depend on who runs it).
The kernel change it measured is NOT in the tree (no effect: DESIGN 5.17): apply tools/probe/matvec_range_order.patch to reproduce."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
for N, rows in ((100_000, 100_000), (100_000, 12_500), (20_000, 20_000), (5_003, 5_003)):
    k = 16
    E, Eb, _ = eng.l2norm(torch.randn(N, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)))
    X = torch.randn(N, k, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
    res = {}
    for rep in range(5):
        for v in (0, 2):
            eng.set_option("matvec_variant", v)
            eng.affinity_matvec(Eb, X, 0, rows)
            eng.profile_begin()
            for _ in range(3): Y = eng.affinity_matvec(Eb, X, 0, rows)
            res.setdefault(v, []).append(eng.profile_end()["affinity_matvec"]["ms"] / 3)
            if rep == 0: res[f"Y{v}"] = Y.clone()
    eng.set_option("matvec_variant", 0)
    fl = 2.0 * rows * N * (192 + k)
    m0, m2 = sorted(res[0])[2], sorted(res[2])[2]
    print(f"N={N} rows={rows}: aligned order {m0:.3f} ms ({fl / m0 / 1e9:.0f} TF)  plain order {m2:.3f} ms ({fl / m2 / 1e9:.0f} TF)  ratio {m0 / m2:.3f}  Y bit-identical: {bool(torch.equal(res['Y0'], res['Y2']))}", flush=True)
