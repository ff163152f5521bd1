#!/usr/bin/env python3
"""The headline step (bench.py --steps 50, no extra legs) with two BINARIES of libsdk_hip.so, alternating child processes on one box: box-to-box
spread (+-3 % between gpurun boxes) is larger than most single changes, so a change to the library is judged here, A against B in the same minutes.
usage: step_two_bin.py <libA> <libB> [rounds]"""
import json, os, statistics, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
libA, libB = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {"A": [], "B": []}
for r in range(rounds):
    for tag, lib in (("A", libA), ("B", libB)):
        env = dict(os.environ, SDK_HIP_LIB=str(Path(lib).resolve()))
        o = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "50", "--no-extras", "--no-cpu-baseline", "--no-affinity-config3"], env=env,
                           capture_output=True, text=True, timeout=400)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(tag, "failed:", o.stderr[-800:]); sys.exit(1)
        j = json.loads(line[-1])
        res[tag].append(j)
        print(tag, lib, "value", j["value"], "ms", j["ms_per_step"], "gemm256 ms", j["kernels"]["conv_gemm256"]["ms"], "clock", j["peaks_used"]["in_kernel_clock_mhz"], flush=True)
a = statistics.median(x["ms_per_step"] for x in res["A"]); b = statistics.median(x["ms_per_step"] for x in res["B"])
ga = statistics.median(x["kernels"]["conv_gemm256"]["ms"] for x in res["A"]); gb = statistics.median(x["kernels"]["conv_gemm256"]["ms"] for x in res["B"])
print(json.dumps({"A": libA, "B": libB, "ms_per_step": {"A": a, "B": b, "B_over_A": round(b / a, 4)}, "conv_gemm256_ms": {"A": ga, "B": gb, "B_over_A": round(gb / ga, 4)},
                  "value": {"A": statistics.median(x["value"] for x in res["A"]), "B": statistics.median(x["value"] for x in res["B"])}}))
