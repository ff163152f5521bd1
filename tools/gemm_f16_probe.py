#!/usr/bin/env python3
"""VERDICT r4 next #3, the GPU half of the decision: conv_gemm256 with fp16 operands (SDK_GEMM_F16, mfma_f32_16x16x32_f16) against the bf16 default
on the same random values - result check, then SUSTAINED time (blocks of back-to-back launches, arms interleaved) and the in-kernel clock:
give-back item 7 says the dtype can change the clock the chip holds."""
import ctypes as C, importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
L = importlib.import_module("speaker-diarization-toolkit_amd._lib")
eng = ops.get_engine(0)
M, T = 201 * 1000, 201
st = torch.cuda.current_stream().cuda_stream


def args(A, W, out, N, Cin, bias, sc, sh, f16):
    g = L.ConvGemmArgs()
    g.A, g.lda, g.W = A.data_ptr(), A.stride(0), W.data_ptr()
    g.C, g.ldc = out.data_ptr(), N
    g.bias, g.scale, g.shift = bias.data_ptr(), sc.data_ptr(), sh.data_ptr()
    g.M, g.N, g.Cin, g.taps, g.dil, g.T = M, N, Cin, 1, 1, T
    g.flags = L.GEMM_RELU | (L.GEMM_F16 if f16 else 0)
    return g


for name, N, Cin, nn in (("1024x1024", 1024, 1024, 100), ("3072x3072", 3072, 3072, 14)):
    Af = torch.randn(M, Cin, device="cuda") * 0.5
    Wf = torch.randn(N, Cin, device="cuda") * 0.03
    bias = torch.randn(N, device="cuda"); sc = torch.rand(N, device="cuda") + 0.5; sh = torch.randn(N, device="cuda")
    arms = {}
    for tag, dt, f16 in (("bf16", torch.bfloat16, False), ("fp16", torch.float16, True)):
        A, W = Af.to(dt), Wf.to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        arms[tag] = (A, W, out, args(A, W, out, N, Cin, bias, sc, sh, f16))
    res = {k: [] for k in arms}
    clk = {}
    for tag, (A, W, out, g) in arms.items():
        L.check(eng.lib.sdk_conv_gemm(eng.ctx, C.byref(g), st), "sdk_conv_gemm")
        torch.cuda.synchronize()
        rows = torch.randint(0, M, (512,), device="cuda")
        want = torch.relu(A[rows].float() @ W.float().T + bias) * sc + sh
        err = (out[rows].float() - want).abs().max().item()
        ulp = 2.0 ** (-8 if tag == "bf16" else -11)
        print(f"  {name} {tag}: max |out - fp32 reference of the same operands| = {err:.3e} (half an output ulp at |v| ~ 4: {4 * ulp / 2:.1e})")
    for b in range(7):
        for tag, (A, W, out, g) in arms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(nn):
                eng.lib.sdk_conv_gemm(eng.ctx, C.byref(g), st)
            e1.record(); e1.synchronize()
            if b:
                res[tag].append(e0.elapsed_time(e1) / nn * 1e3)
    for tag, (A, W, out, g) in arms.items():       # in-kernel clock at the end of a sustained block
        buf = torch.zeros(4096 * 2, dtype=torch.int64, device="cuda")
        for _ in range(nn // 2):
            eng.lib.sdk_conv_gemm(eng.ctx, C.byref(g), st)
        eng.debug_ptr("gemm_clock", buf)
        eng.lib.sdk_conv_gemm(eng.ctx, C.byref(g), st)
        torch.cuda.synchronize()
        eng.debug_ptr("gemm_clock", None)
        t = buf.cpu().numpy().reshape(-1, 2); t = t[(t[:, 0] > 0) & (t[:, 1] > 0)]
        clk[tag] = (float(np.median(t[:, 0] / t[:, 1]) * 100), float(np.median(t[:, 0])) / 1e3)
    fl = 2.0 * M * N * Cin
    print(name, "  ".join(f"{k}: {np.median(v):.1f} us ({fl / np.median(v) / 1e6:.0f} TF, {clk[k][0]:.0f} MHz, {clk[k][1]:.0f} kcycles)" for k, v in res.items()),
          f"  fp16 / bf16 = {np.median(res['fp16']) / np.median(res['bf16']):.4f}", flush=True)
