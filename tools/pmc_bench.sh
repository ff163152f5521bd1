#!/bin/bash
# HBM-side traffic of the bench step per kernel: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmcb; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  # SDK_BIAS_CORRECTION=0: without it the first weight load runs a calibration forward (24 segments) whose launches would be counted as half a pass;
  # the hot path's kernels and bytes are the same either way (only bias VALUES differ)
  SDK_BIAS_CORRECTION=0 SDK_BENCH_PMC=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcb/$c -o b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-affinity-config3 > gpurun_out/pmcb/$c.log 2>&1 || { tail -5 gpurun_out/pmcb/$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, collections
out = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmcb/{c}/*counter_collection.csv")[0]
    rows = list(csv.DictReader(open(f)))
    per_kernel = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] != c: continue
        name = r["Kernel_Name"].split("(")[1].split("::")[-1] if "anonymous" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0]
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].split()[-1]   # template kernels carry a return type
        per_kernel[name].append(float(r["Counter_Value"]))
    for k, v in per_kernel.items():
        n = len(v) // 2          # SDK_BENCH_PMC=1: bench ran exactly 1 warm-up + 1 timed pass of the hot path
        last = v[-n:] if n else v
        out[k][c + "_KB"] = sum(last)
        out[k]["launches"] = len(last)
out["_workload"] = {"segments": 1000, "steps": 1, "warmup": 1, "note": "per-kernel sums over ONE pass of the hot path"}
json.dump(out, open("gpurun_out/pmcb/pmc_bench.json", "w"), indent=1)
for k, v in out.items():
    if not k.startswith("_"): print(k, dict(v))
PY
