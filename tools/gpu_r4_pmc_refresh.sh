#!/bin/bash
# round 4: refresh the counter passes of the step's non-dominant kernels, the mat-vec and the precise GEMM on this round's build
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/pmc_any.sh tools/one_step.py 'res2net_chain|asp_seg|se_apply|conv_gemm_kernel|fbank_tile' step_kernels > gpurun_out/r4_pmc_step.log 2>&1; tail -n 12 gpurun_out/r4_pmc_step.log | cut -c1-250
bash tools/pmc_any.sh tools/one_matvec.py 'affinity_matvec' matvec > gpurun_out/r4_pmc_matvec.log 2>&1; tail -n 3 gpurun_out/r4_pmc_matvec.log | cut -c1-400
bash tools/pmc_any.sh tools/one_step_hp.py 'conv_gemm_hp' precise_gemm > gpurun_out/r4_pmc_hp.log 2>&1; tail -n 4 gpurun_out/r4_pmc_hp.log | cut -c1-400
echo DONE
