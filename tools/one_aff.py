#!/usr/bin/env python3
"""One workload for tools/pmc_any.sh: 1 warm-up + 3 passes of k4 (sdk_affinity_topk, k = 1) at N x P (default config #3: 100000 x 1000)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 1000)
E, Eb, r = eng.l2norm(torch.randn(N, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
Q, Qb, q = eng.l2norm(torch.randn(P, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4)))
qm = q.max().reshape(1)
for _ in range(4):
    eng.affinity_topk(E, Eb, r, Q, Qb, qm, k=1)
torch.cuda.synchronize()
