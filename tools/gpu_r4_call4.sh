#!/bin/bash
# round 4, call 4: the whole GPU suite, the self-launched 2-rank gloo rehearsal of bench.py, PMC traffic refresh, rocprof kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/r4_tests_all.log 2>&1; rc=$?
tail -n 12 gpurun_out/r4_tests_all.log; echo "tests rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
SDK_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4_bench_gloo2.log 2>&1; rc=$?
grep -E '^\{' gpurun_out/r4_bench_gloo2.log | tail -n 1 > gpurun_out/r4_bench_gloo2.json; tail -c 600 gpurun_out/r4_bench_gloo2.log; echo "gloo2 rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
bash tools/pmc_bench.sh > gpurun_out/r4_pmc_bench.log 2>&1; tail -n 14 gpurun_out/r4_pmc_bench.log
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4_rocprof.log 2>&1; echo "rocprof rc=$?"
find gpurun_out/prof -name '*kernel_stats*' | head -n 2
echo DONE
