#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider -k "kblocked or asp_fused or conv_gemm or ecapa_forward" > gpurun_out/r4_kblk_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/r4_kblk_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/option_ab.py h_kblocked > gpurun_out/r4_h_kblocked_ab.log 2>&1; rc=$?; grep -v amdgpu gpurun_out/r4_h_kblocked_ab.log; echo "ab rc=$rc"
