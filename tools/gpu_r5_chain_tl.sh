#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in ${LIBS:-tools/probe/libsdk_hip_r5final.so tools/probe/libsdk_hip_chain1.so tools/probe/libsdk_hip_chainX2.so}; do
  for shape in "1024 1024" "3072 3072"; do
    SDK_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/gemm_timeline.py $shape 2>&1 | grep "per tile" | tee -a gpurun_out/r5_chain_timeline.txt || exit 1
  done
done
