#!/bin/bash
# PMC counters of any kernel family: tools/pmc_any.sh <python script (+ args, quoted)> <kernel-name regex> [tag]
# One rocprofv3 --pmc pass per counter set (only --kernel-trace beside it), the script run once per pass; per-kernel AVERAGES
# per dispatch are printed and written to gpurun_out/pmc_<tag>.json.  FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them
# (gfx950: double FETCH_SIZE for wide streaming reads, MI355X_MICROARCH.md 'HBM').
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
SCRIPT=$1; PAT=$2; TAG=${3:-$PAT}
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -o p -- python3 $SCRIPT > $OUT/s$i.log 2>&1 || { echo "set $i failed"; tail -3 $OUT/s$i.log; }
done
python3 - "$OUT" "$PAT" "$TAG" <<'PY'
import csv, glob, json, re, sys, collections
out_dir, pat, tag = sys.argv[1:4]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{out_dir}/s*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if not re.search(pat, r["Kernel_Name"]): continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].split()[-1]
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches_seen": max(len(v) for v in d.values())} for k, d in per.items()}
json.dump(res, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(x, 1) for c, x in d.items()})
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        print("   of wave cycles: wait_any %.3f  wait_inst %.3f  active_inst %.3f  valu %.3f | mfma busy / (4 * wave cycles / waves-per-simd...) raw: mfma_busy_cycles %.3g busy_cycles %.3g"
              % (d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_ACTIVE_INST_ANY", 0) / wc, d.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                 d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), d.get("SQ_BUSY_CYCLES", 0)))
PY
