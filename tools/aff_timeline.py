#!/usr/bin/env python3
"""In-kernel timeline of the k = 1 affinity coarse kernel (s_memrealtime stamps of wave 0 of every workgroup)."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 1000)
var = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = torch.Generator(device="cuda").manual_seed(3)
E, Eb, re = eng.l2norm(torch.randn(N, 192, device="cuda", generator=g))
Q, Qb, rq = eng.l2norm(torch.randn(P, 192, device="cuda", generator=g))
qm = rq.max().reshape(1)
eng.set_option("affinity_fast_path", 1); eng.set_option("affinity_variant", var)
for _ in range(5): eng.affinity_topk(E, Eb, re, Q, Qb, qm, k=1)
buf = torch.zeros(256 * 64 + 256 * 64, dtype=torch.int64, device="cuda")
eng.debug_ptr("stamps", buf)
eng.affinity_topk(E, Eb, re, Q, Qb, qm, k=1)
torch.cuda.synchronize()
eng.debug_ptr("stamps", None)
allb = buf.cpu().numpy()
t = allb[:256 * 64].reshape(256, 64)
fine = allb[256 * 64:].reshape(256, 2, 32)
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
n = int(t[0, 1])
print(f"{N}x{P} variant {var}: {n - 2} events per workgroup (wg 0)")
print("start of workgroups (us after the first): min %.2f median %.2f max %.2f" % (us(t[:, 0]).min(), np.median(us(t[:, 0])), us(t[:, 0]).max()))
ends = np.array([t[i, int(t[i, 1]) - 1] for i in range(256)])
print("end   of workgroups: min %.2f median %.2f max %.2f" % (us(ends).min(), np.median(us(ends)), us(ends).max()))
clk = np.array([t[i, 63] / max(1, (t[i, int(t[i, 1]) - 1] - t[i, 0])) * 100.0 for i in range(256)])
print("in-kernel clock (MHz): min %.0f median %.0f max %.0f" % (clk.min(), np.median(clk), clk.max()))
for wg in (0, 1, 100, 255):
    k = int(t[wg, 1])
    ev = us(t[wg, 2:k])
    print(f"wg {wg}: start {us(t[wg, 0]):.2f}; events", " ".join(f"{x:.2f}" for x in ev))

st = []
for wg in range(256):
    kk = int(t[wg, 1]); ev = t[wg, 2:kk]
    d = np.diff(ev) / 100.0
    st += [x for x in d if 1.0 < x < 8.0]
print("stage-to-stage intervals (us): median %.2f p10 %.2f p90 %.2f" % (np.median(st), np.percentile(st, 10), np.percentile(st, 90)))
