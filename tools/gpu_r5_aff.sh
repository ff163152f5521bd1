#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "affinity" > gpurun_out/r5_tests_aff.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r5_tests_aff.log | tail -n 8
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/aff_bench.py > gpurun_out/r5_aff_bench.log 2>&1 || { tail gpurun_out/r5_aff_bench.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5_aff_bench.log | tail -n 30
