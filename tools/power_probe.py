#!/usr/bin/env python3
"""Board power and clocks (sysfs hwmon of the amdgpu driver: no HIP context in this process) while a CHILD runs (a) nothing, (b) the config-#2 step back to
back for ~12 s, (c) an HBM-bound sweep (se_apply-like torch add) for ~6 s - the evidence behind 'the GEMMs run at the board's power limit' (DESIGN section 5).
Prints one JSON object: per phase the power cap, mean / max average power, mean shader clock."""
import glob, json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent

def hwmons():
    out = []
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if os.path.exists(h + "/power1_average") or os.path.exists(h + "/power1_input"):
            out.append(h)
    return sorted(out)

def rd(p):
    try:
        return int(open(p).read().split()[0])
    except Exception:
        return None

def sample(h):
    p = rd(h + "/power1_average")
    if p is None:
        p = rd(h + "/power1_input")
    return {"power_W": None if p is None else p / 1e6, "cap_W": (rd(h + "/power1_cap") or 0) / 1e6, "sclk_MHz": (rd(h + "/freq1_input") or 0) / 1e6,
            "mclk_MHz": (rd(h + "/freq2_input") or 0) / 1e6, "temp_C": (rd(h + "/temp1_input") or 0) / 1e3}

CHILD = r'''
import importlib, sys, time, torch
sys.path.insert(0, sys.argv[1])
import bench
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
mode, secs = sys.argv[2], float(sys.argv[3])
if mode == "step":
    eng = ops.get_engine(0)
    pcm = torch.from_numpy(bench.synth_pcm(1000, seed=0)).cuda()
    f = lambda: eng.embed_pcm(pcm)
elif mode in ("gemm", "gemm_zeros", "gemm_k1024", "vendor", "vendor_k1024", "vendor_zeros", "gemm_a_zero", "gemm_w_zero"):
    eng = ops.get_engine(0)
    M = 201000
    N = K = 1024 if mode.endswith("k1024") else 3072
    A = torch.empty(M, K, device="cuda", dtype=torch.bfloat16).normal_(0, 0.5); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    if mode in ("gemm_zeros", "vendor_zeros", "gemm_a_zero"):
        A.zero_()
    if mode in ("gemm_zeros", "vendor_zeros", "gemm_w_zero"):
        W.zero_()
    if mode.startswith("vendor"):
        Wt = W.t(); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        f = lambda: torch.matmul(A, Wt, out=out)             # hipBLASLt, no epilogue: the yardstick of profiles/r05_vendor_sustained.txt
    else:
        f = lambda: eng.conv_gemm(A, W, N, K, relu=True)
else:
    x = torch.randn(201000 * 1024, device="cuda").bfloat16(); y = torch.randn_like(x); z = torch.empty_like(x)
    f = lambda: torch.add(x, y, out=z)
f(); torch.cuda.synchronize()
print("ready", flush=True)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20): f()
    torch.cuda.synchronize(); n += 20
print("done", n, round((time.perf_counter() - t0) / n * 1e3, 4), flush=True)
'''

def phase(h, mode, secs):
    if mode == "idle":
        rows = []
        t0 = time.time()
        while time.time() - t0 < secs:
            rows.append(sample(h)); time.sleep(0.1)
        return rows, None
    p = subprocess.Popen([sys.executable, "-c", CHILD, str(ROOT), mode, str(secs)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    line = p.stdout.readline()
    assert line.startswith("ready"), line
    rows = []
    time.sleep(1.0)                                       # past the ramp
    while p.poll() is None:
        rows.append(sample(h)); time.sleep(0.1)
    tail = p.stdout.read().split()
    return rows[:-3], (float(tail[2]) if len(tail) >= 3 else None)

hs = hwmons()
if not hs:
    print(json.dumps({"error": "no amdgpu hwmon with a power reading is visible to this user", "seen": glob.glob("/sys/class/drm/card*/device/hwmon/*")[:8]})); sys.exit(0)
# the device HIP calls 0 is not necessarily card0: take the hwmon whose power moves under load (measured below); start with all
out = {"hwmons": hs}
def summarise(rows):
    pw = [r["power_W"] for r in rows if r["power_W"] is not None]
    return {"samples": len(rows), "cap_W": rows[0]["cap_W"] if rows else None, "power_mean_W": round(sum(pw) / max(len(pw), 1), 1), "power_max_W": round(max(pw), 1) if pw else None,
            "sclk_mean_MHz": round(sum(r["sclk_MHz"] for r in rows) / max(len(rows), 1)), "mclk_mean_MHz": round(sum(r["mclk_MHz"] for r in rows) / max(len(rows), 1)),
            "temp_max_C": max((r["temp_C"] for r in rows), default=None)}
for mode, secs in (("idle", 2.0), ("step", 12.0), ("gemm", 8.0), ("vendor", 8.0), ("gemm_k1024", 8.0), ("vendor_k1024", 8.0), ("gemm_zeros", 6.0), ("hbm", 6.0), ("idle", 2.0)) if len(sys.argv) < 2 else (("idle", 2.0), ("step", 6.0)) + tuple((m, 6.0) for m in sys.argv[1:]):
    res = {}
    if mode == "idle":
        for h in hs:
            res[os.path.basename(h)] = summarise(phase(h, "idle", secs / len(hs))[0])
    else:
        # sample every hwmon in turn while one child runs
        p_rows = {h: [] for h in hs}
        p = subprocess.Popen([sys.executable, "-c", CHILD, str(ROOT), mode, str(secs)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                             env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        line = p.stdout.readline()
        if not line.startswith("ready"):
            res = {"error": "child did not start: " + line}
        else:
            time.sleep(1.0)
            while p.poll() is None:
                for h in hs:
                    p_rows[h].append(sample(h))
                time.sleep(0.1)
            tail = p.stdout.read().split()
            res = {os.path.basename(h): summarise(rows[:-3]) for h, rows in p_rows.items() if len(rows) > 3}
            res["ms_per_call"] = float(tail[2]) if len(tail) >= 3 else None
    out.setdefault("phases", []).append({"mode": mode, **res})
# keep this process's device only (the box shows every GPU of the host): the hwmon whose power rose most under the step
step = next(p for p in out["phases"] if p["mode"] == "step")
idle = out["phases"][0]
mine = max((k for k in step if k.startswith("hwmon")), key=lambda k: step[k]["power_mean_W"] - idle.get(k, {"power_mean_W": 0})["power_mean_W"])
print(json.dumps({"device_hwmon": mine, "phases": [dict({"mode": p["mode"]}, **p.get(mine, {}), **({"ms_per_call": p["ms_per_call"]} if "ms_per_call" in p else {})) for p in out["phases"]]}))
