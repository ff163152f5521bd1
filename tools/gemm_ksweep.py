#!/usr/bin/env python3
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
g = torch.Generator(device="cuda").manual_seed(0)
M, N = 201000, 1024
if len(sys.argv) > 1: eng.lib.sdk_set_gemm_variant(int(sys.argv[1]))
for Cin in (64, 128, 256, 512, 1024, 2048, 4096):
    A = torch.randn(M, Cin, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    for _ in range(2): eng.conv_gemm(A, W, N, Cin, T=201, bias=bias, scale=bias, shift=bias, relu=True)
    eng.profile_begin()
    for _ in range(5): eng.conv_gemm(A, W, N, Cin, T=201, bias=bias, scale=bias, shift=bias, relu=True)
    pe = eng.profile_end(); p = pe.get("conv_gemm256") or pe["conv_gemm"]
    ms = p["ms"] / 5
    tiles = 786 * 4
    print(f"K={Cin:5d} steps={Cin//64:3d}  {ms:7.3f} ms  per tile-round {ms*1e3*256/tiles*12.28/13:7.2f} us  {2.0*M*N*Cin/ms/1e9:7.0f} TF", flush=True)
