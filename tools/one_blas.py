#!/usr/bin/env python3
"""Yardstick workload for tools/pmc_any.sh: the vendor GEMM (torch.matmul -> hipBLASLt) on the 3072^2 and K = 1024 layer shapes, random operands, 4 launches each."""
import torch
M = 201000
for N, K in ((3072, 3072), (1024, 1024)):
    A = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    for _ in range(4): C = A @ W.t()
    torch.cuda.synchronize()
