#!/bin/bash
# round 5: the accuracy harness on every host path / model family / numerical contract (stand-in voices, random weights: says the pipeline separates
# those voices end to end - nothing about speech), now with precision 2 and with a long recording forced through the chunked ingest ring
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r5_evals_all_paths.txt
echo "# evals/run_eval.py --synthesize on MI355X (round 5, final state)" > $out
run() { name=$1; shift; echo "== $name" >> $out; env "$@" timeout -k 10 300 python evals/run_eval.py --synthesize 2>&1 | grep -v "amdgpu.ids\|SDK_.*WEIGHTS not set" >> $out; echo "rc=$?" >> $out; }
run eval_torch_default SDK_DUMMY=1
run eval_torch_fp16 SDK_PRECISION=2
run eval_torch_precise SDK_PRECISION=1
run eval_torch_default_chunked_ingest SDK_INGEST_CHUNK=65536
run eval_lite SDK_NO_TORCH=1
run eval_lite_chunked_ingest SDK_NO_TORCH=1 SDK_INGEST_CHUNK=65536
run eval_xvector SDK_MODEL=xvector
run eval_xvector_lite SDK_MODEL=xvector SDK_NO_TORCH=1
run eval_xvector_precise SDK_MODEL=xvector SDK_PRECISION=1
run eval_torch_no_bias_correction SDK_BIAS_CORRECTION=0
grep -E "^==|Results|rc=" $out
