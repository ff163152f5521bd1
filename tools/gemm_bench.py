#!/usr/bin/env python3
"""Per-shape conv_gemm timing (HIP events via the in-library profiler), variants interleaved in ONE process."""
import importlib, sys, json
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
shapes_all = [  # (name, M, N, Cin, taps, dil, T)
    ("blk0", 201000, 1024, 128, 5, 1, 201), ("tdnn", 201000, 1024, 1024, 1, 1, 201), ("mfa", 201000, 3072, 3072, 1, 1, 201),
    ("k128", 201000, 1024, 128, 1, 1, 201), ("k4096", 100500, 1024, 4096, 1, 1, 201), ("res2net", 201000, 128, 128, 3, 2, 201),
    ("asp_hidden", 201000, 128, 3072, 1, 1, 201), ("logits", 201000, 3072, 128, 1, 1, 201)]
shapes = [x for x in shapes_all if x[0] in ("tdnn","mfa","k4096")] if len(sys.argv) > 2 else shapes_all
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2"])]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, Cin, taps, dil, T in shapes:
    A = torch.randn(M, Cin, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, taps * Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    res = {}
    for rnd in range(3):
        for v in variants:
            eng.lib.sdk_set_gemm_variant(v)
            for _ in range(2 if rnd == 0 else 0):
                eng.conv_gemm(A, W, N, Cin, taps=taps, dil=dil, T=T, bias=bias, scale=bias, shift=bias, relu=True)
            eng.profile_begin()
            for _ in range(5):
                eng.conv_gemm(A, W, N, Cin, taps=taps, dil=dil, T=T, bias=bias, scale=bias, shift=bias, relu=True)
            pe = eng.profile_end(); p = pe.get("conv_gemm256") or pe["conv_gemm"]
            res.setdefault(v, []).append(p["ms"] / 5)
    fl = 2.0 * M * N * taps * Cin
    print(name, {v: f"{min(t):.3f} ms {fl / min(t) / 1e9:.0f} TF" for v, t in res.items()}, flush=True)
