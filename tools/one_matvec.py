#!/usr/bin/env python3
"""One workload for tools/pmc_any.sh: the config #5 tile kernel (100k x 100k rectified affinity mat-vec, k = 16), 3 launches."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
N, k = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, 16
E, Eb, _ = eng.l2norm(torch.randn(N, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)))
X = torch.randn(N, k, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
for _ in range(3):
    eng.affinity_matvec(Eb, X)
torch.cuda.synchronize()
