#!/usr/bin/env python3
"""The config-#2 forward cut into sequential sub-batches: does a smaller working set (layer tensors inside the 256-MB Infinity Cache) pay for the
GEMM's shorter tile queue?  1000 segments per step as B = 1000 x 1, 500 x 2, 334 x 3, 250 x 4, 200 x 5, 125 x 8; interleaved, 3 passes of ~1 s each."""
import importlib, json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
pcm = torch.from_numpy(bench.synth_pcm(1000, seed=0)).cuda()
def step(nsub):
    return [eng.embed_pcm(c) for c in pcm.chunk(nsub)]
arms = (1, 2, 3, 4, 5, 8)
for n in arms:
    step(n)
torch.cuda.synchronize()
res = {n: [] for n in arms}
for p in range(3):
    for n in arms:
        for _ in range(20): step(n)                      # bring the clock to this arm's steady state
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): step(n)
        torch.cuda.synchronize()
        res[n].append(round((time.perf_counter() - t0) * 10, 4))   # ms per 1000 segments
print(json.dumps({"ms_per_1000_segments_by_sub_batches": res}))
