#!/bin/bash
# conv_gemm256: correctness, then interleaved wall-time A/B (args: variant pairs "a:b ..."), then a short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q --no-header -p no:cacheprovider -x -k "gemm or ecapa or res2net" > gpurun_out/gemm_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/gemm_tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
for pair in "$@"; do
  a=${pair%%:*}; b=${pair##*:}
  timeout -k 10 300 python tools/gemm_ab.py $a $b > gpurun_out/gemm_ab_${a}_${b}.log 2>&1 || { tail gpurun_out/gemm_ab_${a}_${b}.log; exit 1; }
  grep -v amdgpu.ids gpurun_out/gemm_ab_${a}_${b}.log
done
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-affinity-config3 > gpurun_out/bench_sched.log 2>&1 || { tail -n 20 gpurun_out/bench_sched.log; exit 1; }
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_sched.log") if l.startswith("{")][-1])
print("bench:", d["value"], "seg/s", d["ms_per_step"], "ms/step; gemm256", d["kernels"]["conv_gemm256"], "frac", d["roofline"]["frac"], "clock", d["peaks_used"]["in_kernel_clock_mhz"])
PY
