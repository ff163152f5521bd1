#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python tools/aff_debug.py "$@" > gpurun_out/aff_debug.log 2>&1; rc=$?
tail -n 60 gpurun_out/aff_debug.log; exit $rc
