#!/bin/bash
# round 5, call 3: fused column statistics v2 - parity, then A/B of the GEMM shapes (tail on/off) and of the whole step against the round-4 K loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_fp32.py tests/test_gpu_backend_e2e.py -x -q -k "conv_gemm or ecapa or forward or e2e or parity" > gpurun_out/r5_tests_stats.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r5_tests_stats.log | tail -n 25
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/gemm_ab.py 2 8194 > gpurun_out/r5_gemm_ab_stats2.log 2>&1 || { tail gpurun_out/r5_gemm_ab_stats2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5_gemm_ab_stats2.log
timeout -k 10 600 python tools/step_two_bin.py tools/probe/libsdk_hip_r5base.so speaker-diarization-toolkit_amd/libsdk_hip.so 3 > gpurun_out/r5_step_two_bin.log 2>&1 || { tail gpurun_out/r5_step_two_bin.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5_step_two_bin.log
