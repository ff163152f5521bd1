#!/usr/bin/env python3
"""Launch sequence for tools/pmc_gemm.sh: for each layer shape and each gemm_variant, 1 warm + 3 launches of conv_gemm256; the
sequence is written to gpurun_out/gemm_traffic_seq.json so the counter rows (dispatch order) can be attributed."""
import importlib, json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
M, T = 201 * 1000, 201
variants = [int(v) for v in sys.argv[1:]] or [2, 1026, 2050]
shapes = [("tdnn 1024x1024", 1024, 1024, 1, 0), ("mfa 3072x3072", 3072, 3072, 1, 0), ("blk0 k5 128->1024", 1024, 128, 5, 0)]
seq = []
for name, N, Cin, taps, stats in shapes:
    A = (torch.randn(M, Cin, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, taps * Cin, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    for v in variants:
        eng.lib.sdk_set_gemm_variant(v)
        for i in range(4):
            eng.conv_gemm(A, W, N, Cin, taps=taps, T=T, bias=bias, relu=True, stats_mode=stats)
            seq.append({"shape": name, "variant": v, "warm": i == 0, "alg_read_MB": (M * Cin * 2 + N * taps * Cin * 2) / 1e6, "alg_write_MB": M * N * 2 / 1e6})
    torch.cuda.synchronize()
eng.lib.sdk_set_gemm_variant(2)
Path("gpurun_out").mkdir(exist_ok=True)
json.dump(seq, open("gpurun_out/gemm_traffic_seq.json", "w"))
