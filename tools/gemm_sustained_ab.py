#!/usr/bin/env python3
"""Two gemm_variant arms of conv_gemm256 under SUSTAINED load: blocks of `n` back-to-back launches per arm (no host synchronisation inside a block:
the chip stays at the power state of a running forward), arms interleaved, HIP-event time per block.  tools/gemm_ab.py times 3 launches between host
synchronisations - a cooler chip, where saved cycles show as saved time; in the forward they come back as a lower clock.
usage: gemm_sustained_ab.py [variantA variantB] [launches per block] [blocks]"""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
VA, VB = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 8194)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 8
M, T = 201 * 1000, 201
for name, N, Cin, taps, stats in (("tdnn1 1024x1024", 1024, 1024, 1, 0), ("tdnn2 1024x1024 + stats", 1024, 1024, 1, 1), ("mfa 3072x3072 + stats", 3072, 3072, 1, 2)):
    A = (torch.randn(M, Cin, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, taps * Cin, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda"); sc = torch.rand(N, device="cuda") + 0.5; sh = torch.randn(N, device="cuda")
    nn = n if N == 1024 else max(10, n // 7)
    g = eng.conv_gemm_prepared(A, W, N, Cin, taps=taps, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=stats) if hasattr(eng, "conv_gemm_prepared") else None
    def launch():
        if g is not None:
            g()
        else:
            eng.conv_gemm(A, W, N, Cin, taps=taps, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=stats)
    res = {VA: [], VB: []}
    for b in range(blocks + 1):
        for v in (VA, VB):
            eng.lib.sdk_set_gemm_variant(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(nn):
                launch()
            e1.record()
            e1.synchronize()
            if b:                                   # block 0 of each arm warms up
                res[v].append(e0.elapsed_time(e1) / nn * 1e3)
    a, bb = np.median(res[VA]), np.median(res[VB])
    print(f"{name:26s} sustained, {nn} launches per block x {blocks}: variant {VA} {a:8.1f} us   variant {VB} {bb:8.1f} us   ratio {a / bb:.4f}   (blocks {VA}: {[round(x) for x in res[VA]]}, {VB}: {[round(x) for x in res[VB]]})", flush=True)
    del A, W
eng.lib.sdk_set_gemm_variant(2)
