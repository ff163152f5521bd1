#!/usr/bin/env python3
"""k4 coarse pass: boundary penalty of the cost-balanced work split (affinity_boundary_penalty 0 = equal unit counts, 1..4 stages), interleaved."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
for N3, P3 in ((100_000, 1000), (125_000, 10_000), (20_000, 2000)):
    E3, E3b, r3 = eng.l2norm(torch.randn(N3, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
    Q3, Q3b, q3 = eng.l2norm(torch.randn(P3, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4)))
    q3m = q3.max().reshape(1)
    ref, res = None, {}
    for rnd in range(5):
        for pen in (0, 1, 2, 3, 4):
            eng.set_option("affinity_boundary_penalty", pen)
            for _ in range(2): out = eng.affinity_topk(E3, E3b, r3, Q3, Q3b, q3m, k=1, want_count=True)
            torch.cuda.synchronize()
            if ref is None: ref = (out[0].clone(), out[1].clone())
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), f"pen {pen}: answers differ"
            eng.profile_begin()
            for _ in range(10): eng.affinity_topk(E3, E3b, r3, Q3, Q3b, q3m, k=1)
            p = eng.profile_end()
            res.setdefault(pen, []).append(({k: v["ms"] / 10 * 1e3 for k, v in p.items()}, int(out[2].item())))
    for pen, lst in res.items():
        med = {k: sorted(r[0][k] for r in lst)[len(lst) // 2] for k in lst[0][0]}
        tot = sum(med.values())
        print(f"{N3}x{P3} pen {pen}: coarse {med['affinity_coarse']:.1f} us ({2 * N3 * P3 * 192 / med['affinity_coarse'] / 1e6 / 2500:.3f} of peak)  rescore {med['affinity_rescore']:.1f}  rescan {med['affinity_rescan']:.1f}  total {tot:.1f}  total/coarse {tot / med['affinity_coarse']:.2f}  rescanned {lst[0][1]}", flush=True)
eng.set_option("affinity_boundary_penalty", 0)
