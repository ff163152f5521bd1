#!/bin/bash
# the round's standard closing call: whole GPU suite, smoke, full bench line, rocprofv3 kernel stats of the bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r4_tests_final.log 2>&1; rc=$?
tail -n 8 gpurun_out/r4_tests_final.log; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r4_smoke.log 2>&1; rc=$?; tail -n 2 gpurun_out/r4_smoke.log; echo "smoke rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
SECONDS=0
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_final.log 2>&1; rc=$?
echo "bench rc=$rc wall ${SECONDS}s"
grep -E '^\{' gpurun_out/r4_bench_final.log | tail -n 1 > gpurun_out/r4_bench_final.json; head -c 1500 gpurun_out/r4_bench_final.json; echo
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_rocprof.log 2>&1; echo "rocprof rc=$?"
echo DONE
