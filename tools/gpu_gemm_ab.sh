#!/bin/bash
# conv_gemm256: correctness tests, then v3 (overlapped tile boundary) vs v2 A/B in bench runs on the same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py -q --no-header -p no:cacheprovider -x -k "gemm or ecapa or config2 or res2net" > gpurun_out/gemm_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/gemm_tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
for rep in 1 2; do
for v in 2 258; do
  SDK_GEMM_VARIANT=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-affinity-config3 > gpurun_out/bench_g$v.log 2>&1 || { tail -n 20 gpurun_out/bench_g$v.log; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_g$v.log") if l.startswith("{")][-1])
print("variant $v rep $rep:", d["value"], "seg/s", d["ms_per_step"], "ms/step; gemm256", d["kernels"]["conv_gemm256"], "frac", d["roofline"]["frac"], "clock", d["peaks_used"]["in_kernel_clock_mhz"])
PY
done
done
