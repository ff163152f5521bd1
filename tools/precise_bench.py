#!/usr/bin/env python3
"""Default (bf16 operands) vs precise (fp16 hi+lo planes) mode: ms per 1000-segment step and per kernel family."""
import importlib, json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pcm = torch.from_numpy(bench.synth_pcm(B, seed=0)).cuda()
out = {}
for prec in (0, 1):
    eng.set_precision(prec)
    for _ in range(2): eng.embed_pcm(pcm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10 if prec == 0 else 4
    for _ in range(n): eng.embed_pcm(pcm)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    eng.profile_begin(); eng.embed_pcm(pcm); prof = eng.profile_end()
    out[f"precision_{prec}"] = {"ms_per_step": round(ms, 3), "segments_per_s": round(B / ms * 1e3, 1),
                                "kernels_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
                                "kernels_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in prof.items() if v["flops"] and v["ms"] > 0.05}}
eng.set_precision(0)
print(json.dumps(out, indent=1))
