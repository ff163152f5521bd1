#!/bin/bash
# closing evidence of the final tree: one full bench line + the rocprofv3 kernel statistics of the same program
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_final4.log 2>&1; rc=$?
echo "bench rc=$rc wall ${SECONDS}s"
[ $rc -eq 0 ] || exit $rc
grep -E '^\{' gpurun_out/r4_bench_final4.log | tail -n 1 > gpurun_out/r4_bench_final4.json; head -c 400 gpurun_out/r4_bench_final4.json; echo
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/r4_rocprof4.log 2>&1; echo "rocprof rc=$?"
ls gpurun_out/prof | head; rm -f gpurun_out/prof/*kernel_trace.csv gpurun_out/prof/*agent_info.csv
grep -E '^\{' gpurun_out/r4_rocprof4.log | tail -n 1 | head -c 300; echo
echo DONE
