#!/bin/bash
# round 5, call 2: half-tile tail - parity first, then the A/B (one binary: tail on / off; two binaries: round-4 K loop vs this one)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv_gemm or spectral_cluster_survives" -s > gpurun_out/r5_tests_gemm.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r5_tests_gemm.log | tail -n 25
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/gemm_ab.py 2 8194 > gpurun_out/r5_gemm_ab_halftail.log 2>&1 || { tail gpurun_out/r5_gemm_ab_halftail.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5_gemm_ab_halftail.log
timeout -k 10 500 python tools/gemm_two_bin.py tools/probe/libsdk_hip_r5base.so 2 speaker-diarization-toolkit_amd/libsdk_hip.so 8194 3 > gpurun_out/r5_gemm_two_bin.log 2>&1 || { tail gpurun_out/r5_gemm_two_bin.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5_gemm_two_bin.log
