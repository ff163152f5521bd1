#!/usr/bin/env python3
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
for N3, P3 in ((100_000, 1000), (125_000, 10_000)):
    E3, E3b, r3 = eng.l2norm(torch.randn(N3, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
    Q3, Q3b, q3 = eng.l2norm(torch.randn(P3, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4)))
    q3m = q3.max().reshape(1)
    for _ in range(3): eng.affinity_topk(E3, E3b, r3, Q3, Q3b, q3m, k=1)
    eng.profile_begin()
    for _ in range(10): _, _, cnt = eng.affinity_topk(E3, E3b, r3, Q3, Q3b, q3m, k=1, want_count=True)
    p = eng.profile_end()
    tot = sum(v["ms"] for v in p.values()) / 10
    print(N3, P3, {k: round(v["ms"] / 10 * 1e3, 1) for k, v in p.items()}, "us; total", round(tot * 1e3, 1), "us;",
          round(N3 * P3 / tot / 1e6, 1), "Gpairs/s; coarse", round(2 * N3 * P3 * 192 / (p["affinity_coarse"]["ms"] / 10) / 1e9, 1), "TF; rescanned", int(cnt.item()))
