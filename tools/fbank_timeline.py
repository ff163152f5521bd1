#!/usr/bin/env python3
"""In-kernel timeline of fbank_tile_kernel (s_memrealtime stamps of thread 0 of workgroups 0..255)."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
pcm = (torch.randn(1000, 32000, device="cuda") * 3000).to(torch.int16)
for _ in range(3): eng.fbank(pcm)
buf = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
eng.debug_ptr("stamps", buf)
eng.fbank(pcm)
torch.cuda.synchronize()
eng.debug_ptr("stamps", None)
t = buf.cpu().numpy().reshape(256, 16).astype(np.float64) / 100.0
n = int((t[0] > 0).sum())
d = np.diff(t[:, :n], axis=1)
print("events per workgroup:", n, "; lifetime median %.2f us" % np.median(t[:, n - 1] - t[:, 0]))
for k in range(n - 1):
    print(f"phase {k}: median {np.median(d[:, k]):6.2f} us  p10 {np.percentile(d[:, k], 10):6.2f}  p90 {np.percentile(d[:, k], 90):6.2f}")
