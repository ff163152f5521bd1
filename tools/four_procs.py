#!/usr/bin/env python3
"""The reference's process model on one GPU: speaker-process runs up to 4 CLI processes at once (speaker-process:627-629), each building its
backend afresh (base.py:272-293).  Starts N fresh tools/cold_start.py processes AT THE SAME TIME on the one device and reports each one's time to
its first identify row, next to a single process alone - what multi-process sharing of the GPU and of the packed-blob cache costs.
    python tools/four_procs.py [N=4]
"""
import json, os, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
out = {}
with tempfile.TemporaryDirectory(prefix="sdk_cache_") as cache:
    env = dict(os.environ, SDK_CACHE_DIR=cache, HSA_ENABLE_IPC_MODE_LEGACY="0")
    def run(k, extra=()):
        t0 = time.perf_counter()
        ps = [subprocess.Popen([sys.executable, str(ROOT / "tools" / "cold_start.py"), *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(k)]
        res = []
        for p in ps:
            o, _ = p.communicate(timeout=300)
            line = [ln for ln in o.splitlines() if ln.startswith("{")]
            res.append(json.loads(line[-1]) if line else {"error": "no output"})
        return time.perf_counter() - t0, res
    w1, r1 = run(1)                                    # fills the cache
    w1b, r1b = run(1)
    wn, rn = run(n)
    out["alone_empty_cache"] = {"wall_s": round(w1, 3), "time_to_first_row_s": r1[0].get("time_to_first_row_s")}
    out["alone_cache_hit"] = {"wall_s": round(w1b, 3), "time_to_first_row_s": r1b[0].get("time_to_first_row_s")}
    out[f"{n}_at_once_cache_hit"] = {"wall_s": round(wn, 3), "time_to_first_row_s": [r.get("time_to_first_row_s") for r in rn],
                                     "first_enroll_s": [r.get("phases_s", {}).get("first_enroll (code objects, tables, scratch)") for r in rn],
                                     "second_identify_s": [r.get("phases_s", {}).get("second_identify") for r in rn],
                                     "same_model_version": len({r.get("model_version") for r in rn}) == 1}
    wl, rl = run(1, ("--lite",))                       # the torch-free host path (SDK_NO_TORCH=1), same cache
    wln, rln = run(n, ("--lite",))
    out["alone_cache_hit_no_torch"] = {"wall_s": round(wl, 3), "time_to_first_row_s": rl[0].get("time_to_first_row_s")}
    out[f"{n}_at_once_cache_hit_no_torch"] = {"wall_s": round(wln, 3), "time_to_first_row_s": [r.get("time_to_first_row_s") for r in rln],
                                              "torch_imported": [r.get("torch_imported") for r in rln],
                                              "same_model_version": len({r.get("model_version") for r in rln} | {r.get("model_version") for r in rn}) == 1}
print(json.dumps(out))
