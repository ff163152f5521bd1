#!/usr/bin/env python3
"""conv_gemm256 A/B of two gemm_variant arms (default: 2 = half-tile tail on, 8194 = off), interleaved rounds in ONE process on the layer shapes of the forward."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
M, T = 201 * 1000, 201
shapes = [("tdnn 1024x1024", 1024, 1024, 1, 1, 0), ("tdnn2 + stats", 1024, 1024, 1, 1, 1), ("mfa 3072x3072 + stats", 3072, 3072, 1, 1, 2), ("blk0 k5 128->1024", 1024, 128, 5, 1, 0)]
for name, N, Cin, taps, dil, stats in shapes:
    A = (torch.randn(M, Cin, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, taps * Cin, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda"); sc = torch.rand(N, device="cuda") + 0.5; sh = torch.randn(N, device="cuda")
    VA, VB = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 8194)
    res = {VA: [], VB: []}
    outs = {}
    for rnd in range(7):
        for v in (VA, VB):
            eng.lib.sdk_set_gemm_variant(v)
            eng.conv_gemm(A, W, N, Cin, taps=taps, dil=dil, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=stats)
            eng.profile_begin()
            for _ in range(3):
                o = eng.conv_gemm(A, W, N, Cin, taps=taps, dil=dil, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=stats)
            p = eng.profile_end()
            res[v].append(sum(x["ms"] for x in p.values()) / 3)
            outs[v] = o
    same = all(torch.equal(a, b) for a, b in zip(outs[VB] if isinstance(outs[VB], tuple) else (outs[VB],), outs[VA] if isinstance(outs[VA], tuple) else (outs[VA],)) if a is not None)
    m3, m2 = np.median(res[VA]), np.median(res[VB])
    fl = 2.0 * M * N * taps * Cin
    print(f"{name:26s} variant {VA} {m3 * 1e3:8.1f} us ({fl / m3 / 1e9:6.0f} TF)   variant {VB} {m2 * 1e3:8.1f} us ({fl / m2 / 1e9:6.0f} TF)   ratio {m3 / m2:.3f}   outputs identical: {same}", flush=True)
eng.lib.sdk_set_gemm_variant(2)
