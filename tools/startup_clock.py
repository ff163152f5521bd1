#!/usr/bin/env python3
"""is the start-up stall / the slow first steps of a leg the power-state ramp?  A fresh process runs the config-#2 step,
every step synchronised and timed alone, with the in-kernel clock probe of conv_gemm256_kernel (s_memtime / s_memrealtime per workgroup) left
ON, so every step carries (age of the process, wall ms, shader MHz held inside the dominant kernel).  Phases: the first 6 s of the process; a
2-s host pause (GPU idle); 1.5 s more.  Output: one JSON line with every step of the first 400 ms of each phase and every later step slower
than 1.3 x the phase's median."""
import importlib, json, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
t_start = time.perf_counter()
bench = importlib.import_module("bench")
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
t_ctx = time.perf_counter()
pcm = torch.from_numpy(bench.synth_pcm(1000, seed=0)).cuda()
P = torch.from_numpy(bench.unit_rows(100, 192, seed=1)).cuda()
Pn, Pb, rp = eng.l2norm(P)
rpm = rp.max().reshape(1)
eng.desc
buf = torch.zeros(4096 * 2, dtype=torch.int64, device=eng.device)
eng.debug_ptr("gemm_clock", buf)
torch.cuda.synchronize()
t_ready = time.perf_counter()


def step():
    E, Eb, re = eng.embed_pcm(pcm)
    return eng.affinity_topk(E, Eb, re, Pn, Pb, rpm, k=1)


def phase(seconds):
    rows = []
    t_p = time.perf_counter()
    while time.perf_counter() - t_p < seconds:
        buf.zero_()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        t = buf.cpu().numpy().reshape(-1, 2)
        t = t[(t[:, 0] > 0) & (t[:, 1] > 0)]
        mhz = float(np.median(t[:, 0] / t[:, 1]) * 100.0) if len(t) else None   # (the buffer keeps the LAST launch of the step: the 3072^2 layer)
        rows.append((round(t0 - t_ctx, 4), round(t0 - t_p, 4), round(dt, 3), round(mhz, 1) if mhz else None))
    med = float(np.median([r[2] for r in rows]))
    head = [r for r in rows if r[1] < 0.4]
    slow = [r for r in rows if r[1] >= 0.4 and r[2] > 1.3 * med]
    mhzs = [r[3] for r in rows if r[3]]
    return {"steps": len(rows), "median_ms": round(med, 3), "median_mhz": round(float(np.median(mhzs)), 1) if mhzs else None,
            "first_100ms_steps_ms": [r[2] for r in rows if r[1] < 0.1], "first_400ms": head, "later_slow_steps": slow,
            "columns": ["age since context (s)", "age in phase (s)", "step ms", "in-kernel MHz"]}


out = {"setup_s": round(t_ready - t_start, 2), "context_to_ready_s": round(t_ready - t_ctx, 2)}
out["fresh_process_6s"] = phase(6.0)
time.sleep(2.0)
out["after_2s_host_pause"] = phase(1.5)
time.sleep(0.2)
out["after_200ms_host_pause"] = phase(0.8)
print(json.dumps(out))
