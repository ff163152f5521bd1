#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmc/counters.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc/$tag -o aff -- python3 tools/aff_bench.py > gpurun_out/pmc/$tag.log 2>&1 || { tail -5 gpurun_out/pmc/$tag.log; }
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc/*/*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"][:50]
        if "affinity" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in agg.items():
        print(k, {c: round(x) for c, x in v.items()})
PY
