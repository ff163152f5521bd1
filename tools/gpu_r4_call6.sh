#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py -m gpu -q --no-header -p no:cacheprovider -x -k "affinity or config" > gpurun_out/r4_tests6.log 2>&1; rc=$?
tail -n 6 gpurun_out/r4_tests6.log; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 500 python tools/aff_bench.py 100000x1000 > gpurun_out/r4_aff_blocks2.log 2>&1; rc=$?; grep -E "us  coarse|differ|off the" gpurun_out/r4_aff_blocks2.log | tail -n 20
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 200 python tools/aff_timeline.py 100000 1000 8 > gpurun_out/r4_aff_timeline_blocks2.log 2>&1; cat gpurun_out/r4_aff_timeline_blocks2.log
echo DONE
