#!/bin/bash
# conv_gemm256 tile schedules: correctness, fabric traffic (PMC), interleaved wall-time A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q --no-header -p no:cacheprovider -x -k "gemm or ecapa or res2net" > gpurun_out/gemm_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/gemm_tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
bash tools/pmc_gemm.sh 2 1026 2050 || exit 1
timeout -k 10 300 python tools/gemm_ab.py 2 1026 > gpurun_out/gemm_ab_unit.log 2>&1 || { tail gpurun_out/gemm_ab_unit.log; exit 1; }
cat gpurun_out/gemm_ab_unit.log
timeout -k 10 300 python tools/gemm_ab.py 2050 2 > gpurun_out/gemm_ab_pace.log 2>&1 || { tail gpurun_out/gemm_ab_pace.log; exit 1; }
cat gpurun_out/gemm_ab_pace.log
