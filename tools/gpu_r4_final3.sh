#!/bin/bash
# closing validation after the rows_fc fetch-ring change: whole GPU suite + smoke + one bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r4_tests_final3.log 2>&1; rc=$?
tail -n 4 gpurun_out/r4_tests_final3.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/r4_smoke_final3.log 2>&1; rc=$?; tail -n 1 gpurun_out/r4_smoke_final3.log; echo "smoke rc=$rc"
[ $rc -eq 0 ] || exit $rc
SECONDS=0
timeout -k 10 900 python bench.py > gpurun_out/r4_bench_final3.log 2>&1; rc=$?
echo "bench rc=$rc wall ${SECONDS}s"
grep -E '^\{' gpurun_out/r4_bench_final3.log | tail -n 1 > gpurun_out/r4_bench_final3.json; head -c 600 gpurun_out/r4_bench_final3.json; echo
echo DONE
