#!/usr/bin/env python3
"""Is the one-off ~70 ms stall of a fresh process (tools/step_jitter.py) this library's or the platform's?  Pure torch, no libsdk_hip.so: a 5-ms matmul
in a loop from the moment the context exists, every iteration synchronised and timed; prints the iterations that took > 3x the median and their process age."""
import json, time
import torch
t0 = time.perf_counter()
torch.cuda.init()
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize()
t_ctx = time.perf_counter()
rows = []
while time.perf_counter() - t_ctx < 12.0:
    t1 = time.perf_counter()
    for _ in range(8):
        c = a @ b
    torch.cuda.synchronize()
    rows.append((round(t1 - t_ctx, 4), round((time.perf_counter() - t1) * 1e3, 3)))
med = sorted(r[1] for r in rows)[len(rows) // 2]
print(json.dumps({"init_s": round(t_ctx - t0, 3), "iterations": len(rows), "median_ms": med, "first_5": rows[:5],
                  "slow": [r for r in rows if r[1] > 3 * med]}))
