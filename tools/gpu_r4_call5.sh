#!/bin/bash
# round 4, call 5: the block plan of the k4 coarse pass: parity tests, A/B against the range plan, in-kernel timeline; PMC traffic of the bench step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_full_size.py -m gpu -q --no-header -p no:cacheprovider -x -k "affinity or config or spectral" > gpurun_out/r4_tests5.log 2>&1; rc=$?
tail -n 8 gpurun_out/r4_tests5.log; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 500 python tools/aff_bench.py > gpurun_out/r4_aff_blocks.log 2>&1; rc=$?; grep -E "us  coarse|differ|off the" gpurun_out/r4_aff_blocks.log | tail -n 30
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 200 python tools/aff_timeline.py 100000 1000 0 > gpurun_out/r4_aff_timeline_blocks.log 2>&1; cat gpurun_out/r4_aff_timeline_blocks.log
bash tools/pmc_bench.sh > gpurun_out/r4_pmc_bench.log 2>&1; tail -n 16 gpurun_out/r4_pmc_bench.log | cut -c1-160
echo DONE
