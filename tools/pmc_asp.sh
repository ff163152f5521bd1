#!/bin/bash
# PMC counters of the ASP kernel (one rocprofv3 --pmc pass per counter set; no trace domains besides --kernel-trace).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmca; export TMPDIR=/tmp
cat > /tmp/one_asp.py <<'PY'
import importlib, sys, torch
sys.path.insert(0, ".")
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
B, T, C, A = 1000, 201, 3072, 128
g = torch.Generator(device="cuda").manual_seed(0)
h = (torch.randn(B * T, C, device="cuda", generator=g) * 20).to(torch.bfloat16)
ah = torch.tanh(torch.randn(B * T, A, device="cuda", generator=g)).to(torch.bfloat16)
w2 = (torch.randn(C, A, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
b2 = torch.randn(C, device="cuda", generator=g)
for _ in range(3): eng.asp_fused(ah, w2, b2, h, B, T)
torch.cuda.synchronize()
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmca/$tag -o g -- python3 /tmp/one_asp.py > gpurun_out/pmca/$tag.log 2>&1 || tail -3 gpurun_out/pmca/$tag.log
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmca/*/*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in rows:
        if "asp_" not in r["Kernel_Name"]: continue
        per[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print({c: f"{x / max(n[c],1):.4g}" for c, x in per.items()})
PY
