#!/bin/bash
# round 5: counter passes of the final kernels (separate --pmc passes, kernel trace only beside them)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/pmc_bench.sh > gpurun_out/r5_pmc_bench.log 2>&1 || { tail gpurun_out/r5_pmc_bench.log; exit 1; }
cp gpurun_out/pmcb/pmc_bench.json gpurun_out/r05_pmc_bench.json; tail -n 4 gpurun_out/r5_pmc_bench.log
bash tools/pmc_any.sh tools/one_step.py conv_gemm256 gemm256_sq > gpurun_out/r5_pmc_gemm256_sq.log 2>&1 || { tail gpurun_out/r5_pmc_gemm256_sq.log; exit 1; }
tail -n 6 gpurun_out/r5_pmc_gemm256_sq.log
bash tools/pmc_any.sh "tools/one_aff.py 100000 1000" "aff_rowcol_blocks_kernel|aff_rowcol_kernel" aff_cfg3 > gpurun_out/r5_pmc_aff_cfg3.log 2>&1 || { tail gpurun_out/r5_pmc_aff_cfg3.log; exit 1; }
tail -n 4 gpurun_out/r5_pmc_aff_cfg3.log
