#!/usr/bin/env python3
"""Leak / drift check of the plug-in path: one long-lived backend process (the daemon-style use the reference's queue worker implies, speaker-process:627-629 runs
CLI processes, a service would keep the backend), 400 rounds of enroll + identify + verify on recordings of 16 different lengths (3 ... 48 s: the slot sizes,
window counts and batch shapes all vary), 4 speakers in the store.  Reports device memory in use (hipMemGetInfo through torch), torch's allocator, host RSS and
the per-call latency at the start and at the end.
    python tools/soak.py [rounds=400]
"""
import importlib, json, os, resource, sys, tempfile, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
api = importlib.import_module("speaker-diarization-toolkit_amd.plugin_api")
wav = importlib.import_module("speaker-diarization-toolkit_amd.wav")
tmp = Path(tempfile.mkdtemp(prefix="soak_"))
os.environ["SPEAKERS_EMBEDDINGS_DIR"] = str(tmp)
be = api.get_backend("mi355x")
rng = np.random.default_rng(0)
def voice(seconds, f0):
    n = int(16000 * seconds); t = np.arange(n) / 16000.0
    x = sum((0.5 / h) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) for h in range(1, 8)) * (0.6 + 0.4 * np.sin(2 * np.pi * 2.7 * t)) + rng.normal(0, 0.01, n)
    return np.clip(np.round(x / np.abs(x).max() * 0.5 * 32767), -32768, 32767).astype(np.int16)
lengths = [3 + 3 * i for i in range(16)]
files = []
for i, s in enumerate(lengths):
    p = tmp / f"r{i}.wav"; wav.write_wav_s16(p, voice(s, 110 + 20 * (i % 4))); files.append(p)
cands = []
for k in range(4):
    rec = be.enroll_speaker(files[k])
    cands.append({"id": f"spk{k}", "embeddings": {"mi355x": [{"id": f"e{k}", "external_id": rec["external_id"], "model_version": rec["model_version"]}]}})
def rss_mb():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20
def snap():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return {"device_used_MB": round((total - free) / 2**20, 1), "torch_allocated_MB": round(torch.cuda.memory_allocated() / 2**20, 1),
            "torch_reserved_MB": round(torch.cuda.memory_reserved() / 2**20, 1), "host_rss_MB": round(rss_mb(), 1)}
for i in range(16):                                               # every shape once: allocator and scratch reach their steady sizes
    be.identify_speaker(files[i], cands)
snaps, lat, wrong = [dict(snap(), round=0)], [], 0
t_all = time.perf_counter()
for r in range(1, rounds + 1):
    f = files[r % 16]
    t0 = time.perf_counter()
    rows = be.identify_speaker(f, cands)
    v = be.verify_speaker(f, cands[r % 4])
    lat.append(time.perf_counter() - t0)
    if r % 16 < 4 and (not rows or rows[0]["speaker_id"] != f"spk{r % 16}"):
        wrong += 1
    if r % max(50, rounds // 8) == 0:
        rec = be.enroll_speaker(files[r % 16])                   # enrollment writes go on too
        snaps.append(dict(snap(), round=r))
        print(json.dumps(snaps[-1]), flush=True)
a, b = snaps[1] if len(snaps) > 2 else snaps[0], snaps[-1]
print(json.dumps({"rounds": rounds, "seconds": round(time.perf_counter() - t_all, 1), "wrong_top1_on_enrolled_files": wrong,
                  "first": snaps[0], "after_50": a, "last": b, "growth_after_50_MB": {k: round(b[k] - a[k], 1) for k in a if k != "round"},
                  "latency_ms_first_50": round(1e3 * float(np.median(lat[:50])), 2), "latency_ms_last_50": round(1e3 * float(np.median(lat[-50:])), 2)}))
