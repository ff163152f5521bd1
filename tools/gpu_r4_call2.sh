#!/bin/bash
# round 4, call 2: ingest + pack tests, the clock / zero-operand probe of the GEMM, a full bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_backend_e2e.py tests/test_lite.py tests/test_gpu_sentences.py tests/test_xvector.py tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/r4_tests2.log 2>&1; rc=$?
tail -n 15 gpurun_out/r4_tests2.log; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 400 python tools/gemm_clock_ab.py > gpurun_out/r4_gemm_clock_ab.log 2>&1; rc=$?; cat gpurun_out/r4_gemm_clock_ab.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_a.log 2>&1; rc=$?
grep -E '^\{' gpurun_out/r4_bench_a.log | tail -n 1 > gpurun_out/r4_bench_a.json; tail -c 1500 gpurun_out/r4_bench_a.log; echo "bench rc=$rc"
echo DONE
