#!/usr/bin/env python3
"""Res2Net chain: two 4-wave workgroups per CU (default) vs one 8-wave workgroup per CU, interleaved rounds in one process
(kernel time of the three chain launches inside the forward)."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
B, T = 1000, 201
feats = (torch.randn(B * T, 128, device="cuda") * 0.5).bfloat16()
res = {0: [], 1: []}
for rnd in range(6):
    for two in (1, 0):
        eng.set_option("res2net_two_per_cu", two)
        eng.ecapa_forward(feats, B, T)
        eng.profile_begin()
        for _ in range(3): eng.ecapa_forward(feats, B, T)
        p = eng.profile_end()
        res[two].append(p["res2net_chain"]["ms"] / 3)
eng.set_option("res2net_two_per_cu", 1)
a, b = np.median(res[1]), np.median(res[0])
print(f"res2net chain per forward (3 launches): two per CU {a:.4f} ms, one per CU {b:.4f} ms, ratio {a / b:.3f}")
