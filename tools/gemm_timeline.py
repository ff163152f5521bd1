#!/usr/bin/env python3
"""Phase timeline of conv_gemm256 (workgroup 0): per tile, wait for K-step 0 | K loop | epilogue.  $SDK_HIP_LIB selects the binary (two-binary A/B);
usage: gemm_timeline.py [N K [stats_mode]]"""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
M = 201 * 1000
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 1024)
SM = int(sys.argv[3]) if len(sys.argv) > 3 else 0
A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
bias = torch.randn(N, device="cuda"); sc = torch.rand(N, device="cuda") + 0.5; sh = torch.randn(N, device="cuda")
import os
for variant, name in ((2, os.path.basename(os.environ.get("SDK_HIP_LIB", "libsdk_hip.so"))),):
    eng.lib.sdk_set_gemm_variant(variant)
    for _ in range(3): eng.conv_gemm(A, W, N, K, T=201, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=SM)
    buf = torch.zeros(8192, dtype=torch.int64, device="cuda"); stamps = torch.zeros(4096, dtype=torch.int64, device="cuda")
    eng.debug_ptr("gemm_clock", buf); eng.debug_ptr("gemm_stamps", stamps)
    eng.conv_gemm(A, W, N, K, T=201, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=SM)
    torch.cuda.synchronize()
    eng.debug_ptr("gemm_clock", None); eng.debug_ptr("gemm_stamps", None)
    t = buf.cpu().numpy()
    st = stamps.cpu().numpy(); st = st[st > 0].astype(np.float64) / 100.0
    n = len(st) // 3
    st = st[:3 * n].reshape(n, 3)
    wait = st[1:, 0] - st[:-1, 2]; loop = st[:, 1] - st[:, 0]; epi = st[:, 2] - st[:, 1]
    clk = t[:512].reshape(256, 2); mhz = np.median(clk[:, 0] / np.maximum(clk[:, 1], 1)) * 100
    life = clk[:, 1] / 100.0
    ntile = np.array([13 if b < 72 else 12 for b in range(256)])
    print(f"{name}: workgroup lifetimes (us): min {life.min():.1f} median {np.median(life):.1f} max {life.max():.1f}; 13-tile WGs median {np.median(life[ntile == 13]):.1f} max {life[ntile == 13].max():.1f}; "
          f"12-tile WGs median {np.median(life[ntile == 12]):.1f} max {life[ntile == 12].max():.1f}; per XCD class median " + " ".join(f"{np.median(life[x::8]):.0f}" for x in range(8)))
    print(f"{name} N={N} K={K} stats={SM}: {n} tiles by workgroup 0; per tile (us, median): boundary wait {np.median(wait):.2f}, K loop {np.median(loop):.2f} "
          f"({np.median(loop) / (K // 64):.3f} per K-step), epilogue {np.median(epi):.2f}; tile period {np.median(np.diff(st[:, 0])):.2f}; clock {mhz:.0f} MHz")
eng.lib.sdk_set_gemm_variant(2)
