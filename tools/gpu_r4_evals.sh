#!/bin/bash
# round 4: the accuracy harness on every host path / model family / numerical contract (stand-in voices, random weights: says the pipeline separates those voices end to end)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r4_evals_all_paths.txt
echo "# evals/run_eval.py --synthesize on MI355X (round 4, final state): recordings ingested once (pinned staging, windows cut on the device), profiles from the pack where one exists" > $out
run() { name=$1; shift; echo "== $name" >> $out; env "$@" timeout -k 10 300 python evals/run_eval.py --synthesize 2>&1 | grep -v "amdgpu.ids\|SDK_.*WEIGHTS not set" >> $out; echo "rc=$?" >> $out; }
run eval_torch_default SDK_DUMMY=1
run eval_lite SDK_NO_TORCH=1
run eval_torch_precise SDK_PRECISION=1
run eval_xvector SDK_MODEL=xvector
run eval_xvector_lite SDK_MODEL=xvector SDK_NO_TORCH=1
run eval_xvector_precise SDK_MODEL=xvector SDK_PRECISION=1
run eval_torch_no_bias_correction SDK_BIAS_CORRECTION=0
grep -E "^==|Results|rc=" $out
