#!/bin/bash
# A/B of conv_gemm variants in separate bench runs on the same box (kernel-level numbers come from
# the in-process HIP-event profiler, so cross-process variance only affects the wall numbers).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -n 5 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for v in 1 2; do
  SDK_GEMM_VARIANT=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_v$v.log 2>&1 || { tail -n 20 gpurun_out/bench_v$v.log; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_v$v.log") if l.startswith("{")][-1])
print("variant $v", d["value"], "seg/s", d["ms_per_step"], "ms/step; conv_gemm", d["kernels"]["conv_gemm"], "rows_fc", d["kernels"]["rows_fc"])
PY
done
