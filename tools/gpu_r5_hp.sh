#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_precise.py tests/test_xvector.py -x -q > gpurun_out/r5_tests_hp.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r5_tests_hp.log | tail -n 12
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r5_smoke2.log 2>&1 || { tail gpurun_out/r5_smoke2.log; exit 1; }
tail -n 1 gpurun_out/r5_smoke2.log
timeout -k 10 400 python tools/hp_tail_ab.py > gpurun_out/r5_hp_tail_ab.log 2>&1 || { tail gpurun_out/r5_hp_tail_ab.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5_hp_tail_ab.log
