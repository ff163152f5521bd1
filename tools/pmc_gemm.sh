#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmcg; export TMPDIR=/tmp
cat > /tmp/one_gemm.py <<'PY'
import importlib, sys, torch
sys.path.insert(0, ".")
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, Cin) in ((201000, 3072, 3072), (201000, 1024, 1024)):
    A = torch.randn(M, Cin, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    for _ in range(3): eng.conv_gemm(A, W, N, Cin, T=201, relu=True)
torch.cuda.synchronize()
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcg/$tag -o g -- python3 /tmp/one_gemm.py > gpurun_out/pmcg/$tag.log 2>&1 || tail -3 gpurun_out/pmcg/$tag.log
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcg/*/*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if "conv_gemm256" not in r["Kernel_Name"]: continue
        per[(r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X","?"))][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in per.items():
        print(k, {c: f"{x:.4g}" for c, x in v.items()})
PY
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pmcg/SQ_WAVE_CYCLES/*kernel_trace.csv")[0]
for r in csv.DictReader(open(f)):
    if "conv_gemm256" in r["Kernel_Name"]: print(r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
