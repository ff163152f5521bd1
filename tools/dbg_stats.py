import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import test_gpu_fp16 as T16
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
for N in (256, 1024):
    for mode in (1, 2):
        M, T, Cin = 2010, 201, 64
        g = torch.Generator().manual_seed(5)
        A = torch.randn(M, Cin, generator=g)
        Wt = torch.randn(N, Cin, generator=g) * 0.1
        # fp16
        out, st = T16._gemm(eng, A.half().cuda(), Wt.half().cuda(), N, Cin, T=T, stats_mode=mode)
        z = out.cpu().double().reshape(M // T, T, N)
        print("fp16", N, mode, "max |mean err|", float((st.cpu().double()[:, :N] - z.mean(1)).abs().max()), st[0, :10].cpu().numpy().round(4), z.mean(1)[0, :10].numpy().round(4))
        o2 = eng.conv_gemm(A.bfloat16().cuda(), Wt.bfloat16().cuda(), N, Cin, T=T, stats_mode=mode)
        z2 = o2[0].cpu().double().reshape(M // T, T, N)
        print("bf16", N, mode, "max |mean err|", float((o2[3].cpu().double()[:, :N] - z2.mean(1)).abs().max()))
