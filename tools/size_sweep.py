#!/usr/bin/env python3
"""Throughput / latency of the whole path vs batch size (robustness + small-batch latency)."""
import importlib, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
P = torch.randn(100, 192, device="cuda")
Pn, Pb, rp = eng.l2norm(P); rpm = rp.max().reshape(1)
for B in (1, 8, 64, 256, 1000, 4000):
    pcm = torch.randint(-3000, 3000, (B, 32000), dtype=torch.int16, device="cuda")
    def step():
        E, Eb, re = eng.embed_pcm(pcm)
        return eng.affinity_topk(E, Eb, re, Pn, Pb, rpm, k=1)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20 if B <= 256 else 5
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    line = f"B={B:5d}  {dt*1e3:8.3f} ms/step  {B/dt:10.0f} segments/s"
    if B >= 256:
        for ns in (2, 3, 4):
            def step2():
                E, Eb, re = eng.embed_pcm_overlapped(pcm, ns)
                return eng.affinity_topk(E, Eb, re, Pn, Pb, rpm, k=1)
            for _ in range(3): step2()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): step2()
            torch.cuda.synchronize(); d2 = (time.perf_counter() - t0) / n
            line += f"   | {ns} streams {d2*1e3:7.3f} ms {B/d2:8.0f}/s"
    if B <= 256:
        for _ in range(3): eng.embed_pcm_graph(pcm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): eng.embed_pcm_graph(pcm)
        torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / n
        line += f"   | graph replay (embed only) {dg*1e3:7.3f} ms"
    print(line, flush=True)
