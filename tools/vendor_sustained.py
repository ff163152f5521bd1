#!/usr/bin/env python3
"""Yardstick under SUSTAINED load: torch.matmul (hipBLASLt, no epilogue) against conv_gemm256 (bias + ReLU + BN epilogue) on the forward's two GEMM
shapes, blocks of back-to-back launches, arms interleaved (tools/gemm_sustained_ab.py's method).  Not on the product path."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
M, T = 201 * 1000, 201
for name, N, Cin, nn in (("1024x1024", 1024, 1024, 100), ("3072x3072", 3072, 3072, 14)):
    A = (torch.randn(M, Cin, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, Cin, device="cuda") * 0.03).bfloat16()
    Wt = W.t().contiguous()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda"); sc = torch.rand(N, device="cuda") + 0.5; sh = torch.randn(N, device="cuda")
    arms = {"ours (epilogue fused)": lambda: eng.conv_gemm(A, W, N, Cin, T=T, bias=bias, scale=sc, shift=sh, relu=True),
            "vendor A @ W.T": lambda: torch.matmul(A, W.t(), out=out),
            "vendor A @ Wt": lambda: torch.matmul(A, Wt, out=out)}
    res = {k: [] for k in arms}
    for b in range(7):
        for k, fn in arms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(nn):
                fn()
            e1.record(); e1.synchronize()
            if b:
                res[k].append(e0.elapsed_time(e1) / nn * 1e3)
    fl = 2.0 * M * N * Cin
    print(name, "  ".join(f"{k}: {np.median(v):.1f} us ({fl / np.median(v) / 1e6:.0f} TF)" for k, v in res.items()), flush=True)
    del A, W, Wt, out
