#!/usr/bin/env python3
"""Per-step wall time of the config-#2 step from a cold process: where do one-off costs land?  (a) every step synchronised and timed alone,
(b) then bench.py's own pattern (W un-synchronised warm-up steps, fence, K steps, fence) repeated."""
import importlib, json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
bench = importlib.import_module("bench")
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
t_start = time.perf_counter()
eng = ops.get_engine(0)
pcm = torch.from_numpy(bench.synth_pcm(1000, seed=0)).cuda()
P = torch.from_numpy(bench.unit_rows(100, 192, seed=1)).cuda()
Pn, Pb, rp = eng.l2norm(P)
rpm = rp.max().reshape(1)
eng.desc
torch.cuda.synchronize()
t_ready = time.perf_counter()

def step():
    E, Eb, re = eng.embed_pcm(pcm)
    return eng.affinity_topk(E, Eb, re, Pn, Pb, rpm, k=1)

if len(sys.argv) > 1:
    time.sleep(float(sys.argv[1]))                 # does the one-off follow the step count or the clock?
if len(sys.argv) > 2:
    for _ in range(int(sys.argv[2])): eng.l2norm(P)   # ... or the launch count? (extra launches before the first step)
    torch.cuda.synchronize()
single = []
for i in range(30):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); single.append(round((time.perf_counter() - t0) * 1e3, 2))
pattern = []
for rep in range(8):
    time.sleep(0.5 if rep % 2 else 0.0)            # odd repeats start from an idle device
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); pattern.append(round((time.perf_counter() - t0) * 100, 3))
long_run = []
t_l = time.perf_counter()
while time.perf_counter() - t_l < 8.0:             # any later stall?  (age of the process, step time) of every step > 2x the usual
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    if dt > 18.0: long_run.append((round(t0 - t_start, 3), round(dt, 2)))
print(json.dumps({"long_run_8s_slow_steps": long_run}))
print(json.dumps({"setup_s": round(t_ready - t_start, 2), "single_step_ms": single, "w3_k10_ms_per_step": pattern}))
