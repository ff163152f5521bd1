#!/usr/bin/env python3
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
B, T, C, A = 1000, 201, 3072, 128
g = torch.Generator(device="cuda").manual_seed(0)
h = (torch.randn(B * T, C, device="cuda", generator=g) * 20).to(torch.bfloat16)
ah = torch.tanh(torch.randn(B * T, A, device="cuda", generator=g)).to(torch.bfloat16)
w2 = (torch.randn(C, A, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
b2 = torch.randn(C, device="cuda", generator=g)
outs = {}
for per_seg in (1, 0):
    eng.set_option("asp_per_segment", per_seg)
    for _ in range(2): outs[per_seg] = eng.asp_fused(ah, w2, b2, h, B, T)
    eng.profile_begin()
    for _ in range(5): eng.asp_fused(ah, w2, b2, h, B, T)
    p = eng.profile_end()
    print("per-segment workgroups" if per_seg else "(segment, 128 channels) workgroups", {k: round(v["ms"] / 5, 4) for k, v in p.items()}, flush=True)
print("bit-identical:", bool(torch.equal(outs[0], outs[1])), float((outs[0] - outs[1]).abs().max()))
