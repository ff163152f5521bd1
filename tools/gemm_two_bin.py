#!/usr/bin/env python3
"""Two BINARIES of libsdk_hip.so on the forward's GEMM shapes, alternating child processes (SDK_HIP_LIB selects the library): what a change to the
default K loop's code costs or saves when it cannot be an arm of one binary.  usage: gemm_two_bin.py <libA> <variantA> <libB> <variantB> [rounds]
Each child: per shape 2 warm launches + 5 profiled rounds of 3 launches (HIP events on the launch stream), median us, in-kernel clock."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CHILD = r"""
import importlib, json, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
eng.lib.sdk_set_gemm_variant(int(sys.argv[2]))
M, T = 201 * 1000, 201
out = {}
for name, N, Cin, stats in (("k1024", 1024, 1024, 1), ("mfa3072", 3072, 3072, 2)):
    A = (torch.randn(M, Cin, device="cuda") * 0.5).bfloat16()
    W = (torch.randn(N, Cin, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda"); sc = torch.rand(N, device="cuda") + 0.5; sh = torch.randn(N, device="cuda")
    buf = torch.zeros(4096 * 2, dtype=torch.int64, device="cuda")
    for _ in range(6):
        eng.conv_gemm(A, W, N, Cin, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=stats)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        eng.profile_begin()
        for _ in range(3):
            eng.conv_gemm(A, W, N, Cin, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=stats)
        p = eng.profile_end()
        ts.append(sum(x["ms"] for x in p.values()) / 3 * 1e3)
    eng.debug_ptr("gemm_clock", buf)
    eng.conv_gemm(A, W, N, Cin, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=stats)
    torch.cuda.synchronize()
    eng.debug_ptr("gemm_clock", None)
    t = buf.cpu().numpy().reshape(-1, 2); t = t[(t[:, 0] > 0) & (t[:, 1] > 0)]
    out[name] = {"us": round(float(np.median(ts)), 1), "mhz": round(float(np.median(t[:, 0] / t[:, 1]) * 100), 0), "kcycles": round(float(np.median(t[:, 0])) / 1e3, 0)}
    del A, W
print(json.dumps(out))
"""
libA, vA, libB, vB = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
res = {"A": [], "B": []}
for r in range(rounds):
    for tag, lib, v in (("A", libA, vA), ("B", libB, vB)):
        env = dict(os.environ, SDK_HIP_LIB=str(Path(lib).resolve()))
        o = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), v], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(tag, "failed:", o.stderr[-800:]); sys.exit(1)
        res[tag].append(json.loads(line[-1]))
        print(tag, lib, v, line[-1], flush=True)
import statistics
for shape in ("k1024", "mfa3072"):
    a = statistics.median(x[shape]["us"] for x in res["A"]); b = statistics.median(x[shape]["us"] for x in res["B"])
    print(f"{shape}: A {a:.1f} us  B {b:.1f} us  A/B {a / b:.4f}")
