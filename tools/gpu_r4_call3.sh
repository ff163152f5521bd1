#!/bin/bash
# round 4, call 3: x-vector strict modes + lite tests, priority A/Bs of the GEMM and of the affinity coarse pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_xvector.py tests/test_lite.py -m gpu -q --no-header -p no:cacheprovider -x -s > gpurun_out/r4_tests3.log 2>&1; rc=$?
grep -E "deviation|precise mode|passed|failed|Error" gpurun_out/r4_tests3.log | tail -n 12; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python tools/gemm_ab.py 32770 2 > gpurun_out/r4_gemm_prio1.log 2>&1; rc=$?; cat gpurun_out/r4_gemm_prio1.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python tools/gemm_ab.py 65538 2 > gpurun_out/r4_gemm_prio2.log 2>&1; rc=$?; cat gpurun_out/r4_gemm_prio2.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 400 python tools/aff_bench.py > gpurun_out/r4_aff_prio.log 2>&1; rc=$?; tail -n 16 gpurun_out/r4_aff_prio.log
echo DONE
