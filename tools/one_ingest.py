#!/usr/bin/env python3
"""One workload for rocprofv3 --kernel-trace --memory-copy-trace: 2 warm-up + 6 steps of the config-#2 step with its 1000 segments starting in
pageable HOST memory (the ingest path: pinned double-buffered staging, csrc/ingest.hip).  Which engine moves the 64-MB upload?"""
import importlib, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
pcm = bench.synth_pcm(1000, seed=0)
rec = np.ascontiguousarray(pcm).reshape(-1)
tables = {32000: (np.arange(1000, dtype=np.int64) * 32000).astype(np.int32)}
for _ in range(8):
    eng.embed_from_host(rec, tables, step=1000)
torch.cuda.synchronize()
