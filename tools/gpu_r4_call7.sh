#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py -m gpu -q --no-header -p no:cacheprovider -x -k "affinity or config" > gpurun_out/r4_tests7.log 2>&1; rc=$?
tail -n 6 gpurun_out/r4_tests7.log; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/aff_bench.py > gpurun_out/r4_aff_ring.log 2>&1; rc=$?; grep -E "us  coarse|differ|off the" gpurun_out/r4_aff_ring.log | tail -n 30
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 200 python tools/aff_timeline.py 100000 1000 7 > gpurun_out/r4_aff_timeline_ring.log 2>&1; cat gpurun_out/r4_aff_timeline_ring.log
echo DONE
