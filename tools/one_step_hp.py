#!/usr/bin/env python3
"""One workload for tools/pmc_any.sh / rocprofv3: 1 warm-up + 2 passes of the config-#2 hot path in PRECISE mode (1000 segments)."""
import importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
eng.set_precision(1)
pcm = torch.from_numpy(bench.synth_pcm(1000, seed=0)).cuda()
for _ in range(3):
    eng.embed_pcm(pcm)
torch.cuda.synchronize()
