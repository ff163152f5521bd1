#!/usr/bin/env python3
"""The one-off 60-85 ms stall of a fresh process (~1.4 s after its GPU context; bench.py pre-warms past it): is it INSIDE a kernel (the chip runs the
step slowly: clock / power state) or BETWEEN kernels (the queue is not served: runtime / driver)?  Every step of a cold process is timed three ways:
host wall time; device time from HIP events around the step (stream time: kernels + gaps); and the last conv_gemm256 launch's own lifetime in
100 MHz ticks + shader cycles, written by the kernel (debug buffer "gemm_clock").  A step that is slow on the wall and in the events while its
kernel's own lifetime is normal stalled between launches."""
import importlib, json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
bench = importlib.import_module("bench")
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
t_start = time.perf_counter()
eng = ops.get_engine(0)
pcm = torch.from_numpy(bench.synth_pcm(1000, seed=0)).cuda()
eng.desc
torch.cuda.synchronize()
clk = torch.zeros(4096 * 2, dtype=torch.int64, device="cuda")
eng.debug_ptr("gemm_clock", clk)
rows = []
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while time.perf_counter() - t_start < 7.0:
    t0 = time.perf_counter()
    a.record()
    eng.embed_pcm(pcm)
    b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    t = clk.cpu().numpy().reshape(-1, 2)
    t = t[(t[:, 0] > 0) & (t[:, 1] > 0)]
    rows.append((round(t0 - t_start, 3), round(wall, 2), round(a.elapsed_time(b), 2), round(float(np.median(t[:, 1])) / 100.0, 1), round(float(np.median(t[:, 0] / t[:, 1])) * 100.0)))
eng.debug_ptr("gemm_clock", None)
med = np.median([r[1] for r in rows])
slow = [r for r in rows if r[1] > 2 * med]
print(json.dumps({"steps": len(rows), "median_wall_ms": round(float(med), 2), "median_event_ms": round(float(np.median([r[2] for r in rows])), 2),
                  "median_last_gemm_lifetime_us": float(np.median([r[3] for r in rows])), "median_in_kernel_clock_mhz": float(np.median([r[4] for r in rows])),
                  "slow_steps (process age s, wall ms, event ms, last gemm256 lifetime us, its in-kernel clock MHz)": slow,
                  "neighbours_of_first_slow": rows[max(0, rows.index(slow[0]) - 2):rows.index(slow[0]) + 3] if slow else None}))
