#!/usr/bin/env python3
"""In-kernel timeline of res2net_chain4_kernel (two 4-wave workgroups per CU): 9 stamps per conv of wave 0 of workgroups 0..255."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
B, T = 1000, 201
feats = (torch.randn(B * T, 128, device="cuda") * 0.5).bfloat16()
for _ in range(3): eng.ecapa_forward(feats, B, T)
buf = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
eng.debug_ptr("stamps", buf)
eng.ecapa_forward(feats, B, T)
torch.cuda.synchronize()
eng.debug_ptr("stamps", None)
t = buf.cpu().numpy().reshape(256, 64)
names = ["top barrier", "tap0", "tap1", "tap2", "pin + u requests", "barrier (reads done)", "epilogue", "barrier (y done)", "y pass"]
per = {n: [] for n in names}
life = []
for wg in range(256):
    n = int(t[wg, 1])
    ev = t[wg, 2:n].astype(np.float64) / 100.0
    life.append(ev[-1] - t[wg, 0] / 100.0)
    prev = t[wg, 0] / 100.0
    for k, e in enumerate(ev):
        per[names[k % len(names)]].append(e - prev)
        prev = e
print("workgroup lifetime (us): median %.1f min %.1f max %.1f" % (np.median(life), np.min(life), np.max(life)))
for n in names:
    v = np.array(per[n])
    print(f"{n:22s} median {np.median(v):6.2f} us   p10 {np.percentile(v, 10):6.2f}   p90 {np.percentile(v, 90):6.2f}")
