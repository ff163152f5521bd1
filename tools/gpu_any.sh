#!/bin/bash
# run one python tool on the GPU box under a timeout, log under gpurun_out/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
name=$(basename "$1" .py)
timeout -k 10 ${TMO:-300} python "$@" > "gpurun_out/$name.log" 2>&1; rc=$?
tail -n ${TAILN:-60} "gpurun_out/$name.log"; exit $rc
