#!/bin/bash
# conv_gemm256 fabric traffic per tile schedule: separate --pmc passes for FETCH_SIZE and WRITE_SIZE over tools/gemm_traffic.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmcg; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcg/$c -o g -- python3 tools/gemm_traffic.py "$@" > gpurun_out/pmcg/$c.log 2>&1 || { tail -5 gpurun_out/pmcg/$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, collections
seq = json.load(open("gpurun_out/gemm_traffic_seq.json"))
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmcg/{c}/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "conv_gemm256" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    assert len(rows) == len(seq), (len(rows), len(seq))
    vals[c] = [float(r["Counter_Value"]) for r in rows]
agg = collections.OrderedDict()
for i, s in enumerate(seq):
    if s["warm"]: continue
    k = (s["shape"], s["variant"])
    a = agg.setdefault(k, {"fetch_kb": [], "write_kb": [], "alg_read_MB": s["alg_read_MB"], "alg_write_MB": s["alg_write_MB"]})
    a["fetch_kb"].append(vals["FETCH_SIZE"][i]); a["write_kb"].append(vals["WRITE_SIZE"][i])
out = []
for (shape, v), a in agg.items():
    # gfx950: FETCH_SIZE counts 64-B units as if they were 32-B (x2); both counters are in KiB-like units of 1024 B
    rd = 2 * sum(a["fetch_kb"]) / len(a["fetch_kb"]) * 1024 / 1e6
    wr = sum(a["write_kb"]) / len(a["write_kb"]) * 1024 / 1e6
    rec = {"shape": shape, "variant": v, "read_MB": round(rd, 1), "write_MB": round(wr, 1), "alg_read_MB": round(a["alg_read_MB"], 1), "alg_write_MB": round(a["alg_write_MB"], 1),
           "read_over_alg": round(rd / a["alg_read_MB"], 2), "total_over_alg": round((rd + wr) / (a["alg_read_MB"] + a["alg_write_MB"]), 2)}
    out.append(rec); print(rec)
json.dump(out, open("gpurun_out/pmcg/pmc_gemm.json", "w"), indent=1)
PY
