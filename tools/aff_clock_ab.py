#!/usr/bin/env python3
"""Is the k4 coarse pass (aff_rowcol_kernel) held back by stalls or by the clock the chip grants?  Same launch, same instruction stream, on random unit
rows and on CONSTANT rows (every element 1/sqrt(192): no switching in the matrix datapath), interleaved in one process; in-kernel clock from the
kernel's own stamps (debug buffer "stamps": shader cycles / 100 MHz ticks of every workgroup's lifetime)."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
for N, P in ((100_000, 1000), (125_000, 10_000)):
    g = torch.Generator(device="cuda").manual_seed(3)
    Er, Erb, rr = eng.l2norm(torch.randn(N, 192, device="cuda", generator=g))
    Qr, Qrb, qr = eng.l2norm(torch.randn(P, 192, device="cuda", generator=g))
    Ec, Ecb, rc = eng.l2norm(torch.ones(N, 192, device="cuda"))
    Qc, Qcb, qc = eng.l2norm(torch.ones(P, 192, device="cuda"))
    arms = {"random": (Er, Erb, rr, Qr, Qrb, qr.max().reshape(1)), "constant": (Ec, Ecb, rc, Qc, Qcb, qc.max().reshape(1))}
    res = {k: {"us": [], "mhz": []} for k in arms}
    buf = torch.zeros(256 * 64 + 256 * 64, dtype=torch.int64, device="cuda")
    for rnd in range(7):
        for name, a in arms.items():
            eng.affinity_topk(*a, k=1)
            eng.profile_begin()
            for _ in range(5):
                eng.affinity_topk(*a, k=1)
            p = eng.profile_end()
            res[name]["us"].append(p["affinity_coarse"]["ms"] / 5 * 1e3)
            buf.zero_()
            eng.debug_ptr("stamps", buf)
            eng.affinity_topk(*a, k=1)
            torch.cuda.synchronize()
            eng.debug_ptr("stamps", None)
            t = buf.cpu().numpy()[:256 * 64].reshape(256, 64)
            clk = [t[i, 63] / max(1, t[i, int(t[i, 1]) - 1] - t[i, 0]) * 100.0 for i in range(256) if t[i, 1] > 2]
            res[name]["mhz"].append(float(np.median(clk)))
    fl = 2.0 * N * P * 192
    for name in arms:
        us, mhz = np.median(res[name]["us"]), np.median(res[name]["mhz"])
        print(f"{N}x{P} {name:9s} coarse {us:7.1f} us  {fl / us / 1e6:6.0f} TF  in-kernel clock {mhz:5.0f} MHz  cycles {us * mhz:9.0f}", flush=True)
