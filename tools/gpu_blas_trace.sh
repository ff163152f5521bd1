#!/bin/bash
# which kernels the vendor GEMM runs on the conv_gemm256 shapes (yardstick only): kernel-trace stats of tools/blas_ref.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/blas; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/blas -o b -- python3 tools/blas_ref.py > gpurun_out/blas/run.log 2>&1 || { tail -5 gpurun_out/blas/run.log; exit 1; }
grep -v amdgpu.ids gpurun_out/blas/run.log | tail -5
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/blas/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:200], r["Calls"], r["AverageNs"])
f = glob.glob("gpurun_out/blas/*kernel_trace.csv")[0]
seen = set()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:60], r["Workgroup_Size"], r["Grid_Size"], r["LDS_Block_Size"], r.get("VGPR_Count"), r.get("Accum_VGPR_Count"))
    if k not in seen: seen.add(k); print(k)
PY
