#!/bin/bash
# affinity A/B + the affinity parity tests, one gpurun call
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 420 python -m pytest tests/test_gpu_kernels.py -q --no-header -p no:cacheprovider -x -k "affinity and not matvec" > gpurun_out/aff_tests.log 2>&1; rc=$?
tail -n 15 gpurun_out/aff_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "ABORT tests hung"; exit $rc; fi
timeout -k 10 420 python tools/aff_bench.py "$@" > gpurun_out/aff_bench.log 2>&1; rc=$?
cat gpurun_out/aff_bench.log | tail -n 40
exit $rc
