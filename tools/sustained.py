#!/usr/bin/env python3
"""The config-#2 step for ~20 s without a pause (thermal / DVFS behaviour of the headline): throughput per 2-s slice."""
import importlib, json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
pcm = torch.from_numpy(bench.synth_pcm(1000, seed=0)).cuda()
P = torch.from_numpy(bench.unit_rows(100, 192, seed=1)).cuda()
Pn, Pb, rp = eng.l2norm(P); rpm = rp.max().reshape(1)
def step():
    E, Eb, re = eng.embed_pcm(pcm)
    return eng.affinity_topk(E, Eb, re, Pn, Pb, rpm, k=1)
for _ in range(3): step()
torch.cuda.synchronize()
slices, t_all = [], time.perf_counter()
for s in range(10):
    t0 = time.perf_counter()
    for _ in range(230): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    slices.append(round(230 * 1000 / dt, 1))
print(json.dumps({"steps": 2300, "seconds": round(time.perf_counter() - t_all, 2), "segment_embeddings_per_s_by_slice": slices,
                  "min": min(slices), "max": max(slices), "first_over_last": round(slices[0] / slices[-1], 4)}))
