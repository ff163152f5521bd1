#!/usr/bin/env python3
"""Is conv_gemm256 held back by stalls or by the clock the chip grants (power)?  For the K = 1024 layer shape and the 3072^2 layer, interleaved in
ONE process: variant 2 (default) and 258 (v3: overlapped tile boundary, 6-7 % fewer cycles per tile) on RANDOM operands, and variant 2 on
ALL-ZERO operands (same instruction stream, same stalls, far less switching in the matrix datapath).  Per arm: median time and the median
in-kernel clock (s_memtime / s_memrealtime per workgroup, debug buffer "gemm_clock")."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
M, T = 201 * 1000, 201
clk = torch.zeros(4096 * 2, dtype=torch.int64, device="cuda")
for name, N, Cin in (("tdnn 1024x1024", 1024, 1024), ("mfa 3072x3072", 3072, 3072)):
    A = (torch.randn(M, Cin, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, Cin, device="cuda") * 0.03).bfloat16()
    A0, W0 = torch.zeros_like(A), torch.zeros_like(W)
    Ap = torch.relu(A.float()).bfloat16()                       # post-ReLU-like: half the elements exactly zero, the rest positive
    bias = torch.randn(N, device="cuda"); sc = torch.rand(N, device="cuda") + 0.5; sh = torch.randn(N, device="cuda")
    arms = [("v2 random", 2, A, W), ("v3 random", 258, A, W), ("v2 zeros", 2, A0, W0), ("v2 relu-like A", 2, Ap, W),
            ("v2 A random, W zero", 2, A, W0), ("v2 A zero, W random", 2, A0, W)]     # all products zero either way: what is left is the cost of MOVING one random operand
    res = {a[0]: {"us": [], "mhz": []} for a in arms}
    for rnd in range(7):
        for label, var, a, w in arms:
            eng.lib.sdk_set_gemm_variant(var)
            eng.conv_gemm(a, w, N, Cin, T=T, bias=bias, scale=sc, shift=sh, relu=True)
            eng.profile_begin()
            for _ in range(3):
                eng.conv_gemm(a, w, N, Cin, T=T, bias=bias, scale=sc, shift=sh, relu=True)
            p = eng.profile_end()
            res[label]["us"].append(sum(x["ms"] for x in p.values()) / 3 * 1e3)
            clk.zero_()
            eng.debug_ptr("gemm_clock", clk)
            eng.conv_gemm(a, w, N, Cin, T=T, bias=bias, scale=sc, shift=sh, relu=True)
            torch.cuda.synchronize()
            eng.debug_ptr("gemm_clock", None)
            t = clk.cpu().numpy().reshape(-1, 2)
            t = t[(t[:, 0] > 0) & (t[:, 1] > 0)]
            res[label]["mhz"].append(float(np.median(t[:, 0] / t[:, 1]) * 100.0))
    fl = 2.0 * M * N * Cin
    base = np.median(res["v2 random"]["us"])
    for label, _, _, _ in arms:
        us, mhz = np.median(res[label]["us"]), np.median(res[label]["mhz"])
        print(f"{name:16s} {label:16s} {us:8.1f} us  {fl / us / 1e6:6.0f} TF  x{us / base:.3f} of v2 random   in-kernel clock {mhz:6.0f} MHz   cycles {us * mhz:9.0f}", flush=True)
eng.lib.sdk_set_gemm_variant(2)
