#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fp16.py -x -q -s > gpurun_out/r5_tests_fp16.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r5_tests_fp16.log | tail -n 40
exit $rc
