#!/bin/bash
# round 4, call 1: the re-parametrised Backend-level tests, the K-blocked source-address timing probe, the SQ counter pass of the default
# conv_gemm256 kernel and the PMC passes of aff_rowcol at both shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_backend_e2e.py tests/test_lite.py tests/test_gpu_sentences.py tests/test_c_host.py tests/test_gpu_bias_correction.py -m gpu -q --no-header -p no:cacheprovider -x -s > gpurun_out/r4_tests1.log 2>&1; rc=$?
tail -n 15 gpurun_out/r4_tests1.log; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python tools/gemm_ab.py 8194 2 > gpurun_out/r4_kblock_A.log 2>&1; rc=$?; cat gpurun_out/r4_kblock_A.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python tools/gemm_ab.py 24578 2 > gpurun_out/r4_kblock_AW.log 2>&1; rc=$?; cat gpurun_out/r4_kblock_AW.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
bash tools/pmc_any.sh tools/one_step.py 'conv_gemm256' gemm256_sq > gpurun_out/r4_pmc_gemm256.log 2>&1; tail -n 12 gpurun_out/r4_pmc_gemm256.log
bash tools/pmc_any.sh "tools/one_aff.py 100000 1000" 'aff_rowcol' aff_cfg3 > gpurun_out/r4_pmc_aff3.log 2>&1; tail -n 8 gpurun_out/r4_pmc_aff3.log
bash tools/pmc_any.sh "tools/one_aff.py 125000 10000" 'aff_rowcol' aff_cfg4 > gpurun_out/r4_pmc_aff4.log 2>&1; tail -n 8 gpurun_out/r4_pmc_aff4.log
echo DONE
