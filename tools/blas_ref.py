#!/usr/bin/env python3
"""Yardstick only: what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the conv_gemm256 shapes."""
import torch, sys
M = 201000
for N, K in ((1024, 640), (1024, 1024), (3072, 3072), (1024, 4096)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    for _ in range(3): C = A @ W.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): C = A @ W.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"N={N} K={K}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.0f} TF", flush=True)
