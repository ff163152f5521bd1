#!/usr/bin/env python3
"""Where a K-step of conv_gemm256 waits (workgroup 0, thread 0; gemm_variant 2050 = v2 + K-step stamps): per K-step, time from the
end of sub-phase P2 to 'own DMA of the next step landed' (vmcnt), from there to 'every wave arrived' (barrier), and the rest."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
M = 201 * 1000
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2050     # 2050 = default kernel + K-step stamps
for N, K in ((1024, 1024), (3072, 3072)):
    A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    eng.lib.sdk_set_gemm_variant(variant)
    for _ in range(3): eng.conv_gemm(A, W, N, K, T=201, bias=bias, relu=True)
    buf = torch.zeros(8192, dtype=torch.int64, device="cuda"); stamps = torch.zeros(4096, dtype=torch.int64, device="cuda")
    eng.debug_ptr("gemm_clock", buf); eng.debug_ptr("gemm_stamps", stamps)
    eng.conv_gemm(A, W, N, K, T=201, bias=bias, relu=True)
    torch.cuda.synchronize()
    eng.debug_ptr("gemm_clock", None); eng.debug_ptr("gemm_stamps", None)
    eng.lib.sdk_set_gemm_variant(2)
    t = buf.cpu().numpy()
    st = stamps.cpu().numpy(); st = st[st > 0].astype(np.float64) / 100.0
    nk = K // 64
    per_tile = 1 + 3 * (nk - 1) + 1 + 1          # [K-step 0 landed] + 3 per K-step but the last + [K loop end] + [tile end]
    nt = len(st) // per_tile
    dma, bar, rest = [], [], []
    for k in range(nt):
        e = st[k * per_tile:(k + 1) * per_tile]
        ks = e[1:1 + 3 * (nk - 1)].reshape(nk - 1, 3)
        dma += list(ks[:, 1] - ks[:, 0]); bar += list(ks[:, 2] - ks[:, 1])
        rest += list(np.diff(ks[:, 0]))
    clk = t[:512].reshape(256, 2); mhz = np.median(clk[:, 0] / np.maximum(clk[:, 1], 1)) * 100
    print(f"variant {variant} N={N} K={K}: {nt} tiles; K-step period median {np.median(rest):.2f} us (p90 {np.percentile(rest, 90):.2f}); wait for own DMA median {np.median(dma):.2f} (mean {np.mean(dma):.2f}, p90 {np.percentile(dma, 90):.2f}); "
          f"barrier median {np.median(bar):.2f} (mean {np.mean(bar):.2f}, p90 {np.percentile(bar, 90):.2f}); clock {mhz:.0f} MHz; MFMA time per K-step at that clock {2 * 64 * 16 / mhz:.2f} us")
