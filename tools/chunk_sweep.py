#!/usr/bin/env python3
"""Does running the 1000-segment step as end-to-end chunks (activations small enough to stay in the 256-MB
Infinity Cache) beat one big pass?  Chunk sizes chosen so M = 201*B is just under a multiple of 256 rows * 64 tiles."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
B = 1000
pcm = torch.randint(-3000, 3000, (B, 32000), dtype=torch.int16, device="cuda")
for cb in (1000, 500, 334, 326, 250, 163, 125):
    chunks = [pcm[i:i + cb] for i in range(0, B, cb)]
    def step():
        return [eng.embed_pcm(c)[0] for c in chunks]
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"chunk={cb:5d} x{len(chunks)}  {dt*1e3:8.3f} ms/1000 segments  {B/dt:10.0f} segments/s", flush=True)
