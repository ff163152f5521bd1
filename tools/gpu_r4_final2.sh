#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r4_tests_final2.log 2>&1; rc=$?
tail -n 4 gpurun_out/r4_tests_final2.log; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python tools/gemm_ab.py 258 2 > gpurun_out/r4_gemm_ab_restored.log 2>&1; grep -v amdgpu gpurun_out/r4_gemm_ab_restored.log
timeout -k 10 400 python tools/aff_bench.py 100000x1000 > gpurun_out/r4_aff_final.log 2>&1; grep -E "us  coarse" gpurun_out/r4_aff_final.log
timeout -k 10 120 python tools/stall_probe2.py > gpurun_out/r4_stall_probe2.log 2>&1; tail -n 1 gpurun_out/r4_stall_probe2.log | cut -c1-1500
SECONDS=0
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_final2.log 2>&1; rc=$?
echo "bench rc=$rc wall ${SECONDS}s"
grep -E '^\{' gpurun_out/r4_bench_final2.log | tail -n 1 > gpurun_out/r4_bench_final2.json; head -c 700 gpurun_out/r4_bench_final2.json; echo
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_rocprof.log 2>&1; echo "rocprof rc=$?"
echo DONE
