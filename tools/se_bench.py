#!/usr/bin/env python3
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
B, T, C, Cse = 1000, 201, 1024, 128
g = torch.Generator(device="cuda").manual_seed(0)
z = torch.randn(B * T, C, device="cuda", generator=g).to(torch.bfloat16)
x = torch.randn(B * T, 3072, device="cuda", generator=g).to(torch.bfloat16)[:, :C]
w1t = torch.randn(C, Cse, device="cuda", generator=g) / 32
w2t = torch.randn(Cse, C, device="cuda", generator=g) / 11
b1 = torch.zeros(Cse, device="cuda"); b2 = torch.zeros(C, device="cuda")
for _ in range(2): eng.se_gate_residual(z, x, w1t, b1, w2t, b2, B, T)
eng.profile_begin()
for _ in range(5): eng.se_gate_residual(z, x, w1t, b1, w2t, b2, B, T)
p = eng.profile_end()
ms = p["se_gate"]["ms"] / 5
print("se_gate", round(ms, 4), "ms", round(4 * B * T * C * 2 / ms / 1e6, 1), "GB/s (z twice + x + out)")
h = torch.randn(B * T, 3072, device="cuda", generator=g).to(torch.bfloat16)
eng.profile_begin()
for _ in range(5): eng.asp_stats(h, B, T)
p = eng.profile_end()
print("asp_stats", round(p["asp_stats"]["ms"] / 5, 4), "ms", round(B * T * 3072 * 2 / (p["asp_stats"]["ms"] / 5) / 1e6, 1), "GB/s")
