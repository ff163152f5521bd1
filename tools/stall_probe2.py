#!/usr/bin/env python3
"""The one-off 60-85 ms stall of a fresh process (~1.4 s after its GPU context; bench.py pre-warms past it): is it INSIDE a kernel (the chip runs
slowly: clock / power state) or BETWEEN kernels (the queue is not served: runtime / driver)?  From the moment the context exists, ONE conv_gemm256
launch per iteration (~0.25 ms, M = 256 x 201), every iteration synchronised and timed three ways: host wall time, HIP events around the launch
(stream time), and the kernel's OWN lifetime in 100 MHz ticks + shader cycles (debug buffer "gemm_clock").  An iteration that is slow on the wall and
in the events while the kernel's own lifetime is normal stalled outside the kernel."""
import importlib, json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
t_start = time.perf_counter()
eng = ops.Engine(0, bias_correction=False)
M, T = 201 * 256, 201
A = (torch.randn(M, 1024, device="cuda") * 0.5).bfloat16()
W = (torch.randn(1024, 1024, device="cuda") * 0.03).bfloat16()
torch.cuda.synchronize()
t_ctx = time.perf_counter()
clk = torch.zeros(4096 * 2, dtype=torch.int64, device="cuda")
eng.debug_ptr("gemm_clock", clk)
rows = []
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while time.perf_counter() - t_ctx < 6.0:
    t0 = time.perf_counter()
    a.record()
    eng.conv_gemm(A, W, 1024, 1024, T=T, relu=True)
    b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    t = clk[:512].cpu().numpy().reshape(-1, 2)
    t = t[(t[:, 0] > 0) & (t[:, 1] > 0)]
    rows.append((round(t0 - t_ctx, 4), round(wall, 3), round(a.elapsed_time(b), 3), round(float(np.max(t[:, 1])) / 100.0, 1), round(float(np.median(t[:, 0] / t[:, 1])) * 100.0)))
eng.debug_ptr("gemm_clock", None)
med = float(np.median([r[1] for r in rows]))
slow = [r for r in rows if r[1] > 5 * med]
i0 = rows.index(slow[0]) if slow else 0
print(json.dumps({"context_ready_after_s": round(t_ctx - t_start, 2), "iterations": len(rows), "median_wall_ms": round(med, 3),
                  "median_event_ms": round(float(np.median([r[2] for r in rows])), 3), "median_kernel_lifetime_us": float(np.median([r[3] for r in rows])),
                  "slow_iterations (s after context, wall ms, event ms, the kernel's own lifetime us, its in-kernel clock MHz)": slow[:8],
                  "neighbours_of_first_slow": rows[max(0, i0 - 2):i0 + 3] if slow else None}))
