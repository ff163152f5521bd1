#!/bin/bash
# round 5, late: the chained MFMA order of conv_gemm256 (each accumulator's two products back to back) - GEMM tests, timelines, then the step with two binaries
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16.py -m gpu -x -q -k "gemm or forward or ecapa" > gpurun_out/r5_chain_tests.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r5_chain_tests.log | tail -n 4
[ $rc -ne 0 ] && exit $rc
LIBS="tools/probe/libsdk_hip_r5final.so speaker-diarization-toolkit_amd/libsdk_hip.so" bash tools/gpu_r5_chain_tl.sh || exit 1
timeout -k 10 600 python tools/step_two_bin.py tools/probe/libsdk_hip_r5final.so speaker-diarization-toolkit_amd/libsdk_hip.so 3 > gpurun_out/r5_chain_step.txt 2>&1 || { tail gpurun_out/r5_chain_step.txt; exit 1; }
tail -n 1 gpurun_out/r5_chain_step.txt
