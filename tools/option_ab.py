#!/usr/bin/env python3
"""A context option (sdk_set_option NAME) on vs off on the config-#2 forward (1000 segments, T = 201), interleaved rounds in ONE process:
per-kernel times + the whole forward; embeddings compared bit for bit.  usage: option_ab.py NAME [restore_value]"""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
WP = importlib.import_module("speaker-diarization-toolkit_amd.weights_pack")
eng = ops.get_engine(0)
OPT = sys.argv[1]
RESTORE = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, T = 1000, 201
g = torch.Generator(device="cuda").manual_seed(0)
f = torch.zeros(B * T, WP.N_MELS_PADDED, dtype=torch.bfloat16, device="cuda")
f[:, :80] = (torch.randn(B * T, 80, device="cuda", generator=g) * 3).to(torch.bfloat16)
res = {0: [], 1: []}
outs = {}
for rnd in range(7):
    for v in (1, 0):
        eng.set_option(OPT, v)
        eng.ecapa_forward(f, B, T)
        eng.profile_begin()
        for _ in range(3):
            outs[v] = eng.ecapa_forward(f, B, T).clone()
        p = eng.profile_end()
        res[v].append({k: x["ms"] / 3 for k, x in p.items()})
eng.set_option(OPT, RESTORE)
keys = sorted(res[1][0])
print(f"{'kernel':22s} {'  option=1 ms':>13s} {'  option=0 ms':>13s}  ratio")
tot = {0: 0.0, 1: 0.0}
for k in keys:
    a, b = np.median([r[k] for r in res[1]]), np.median([r.get(k, 0.0) for r in res[0]])
    tot[1] += a; tot[0] += b
    if abs(a - b) > 0.002 or k in ("conv_gemm", "asp_fused", "conv_gemm256", "se_gate"):
        print(f"{k:22s} {a:13.4f} {b:13.4f}  {a / b if b else float('nan'):.3f}")
print(f"{'forward (sum)':22s} {tot[1]:13.4f} {tot[0]:13.4f}  {tot[1] / tot[0]:.4f}")
print("embeddings bit-identical:", bool(torch.equal(outs[0], outs[1])))
