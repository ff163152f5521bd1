#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprof summary.  A step that is killed or
# times out (rc 124/137) stops the script: no further GPU step runs after a hang.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
stage() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name ===" | tee -a gpurun_out/run.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a gpurun_out/run.log
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "ABORT: $name hung/killed" | tee -a gpurun_out/run.log; exit $rc; fi
  return $rc
}
: > gpurun_out/run.log
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = test ]; then
  stage pytest_gpu 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider ${PYTEST_ARGS--x}
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  stage smoke 300 python __graft_entry__.py smoke
  stage bench 600 python bench.py --steps ${STEPS:-10} --warmup 3
  grep -E '^\{' gpurun_out/bench.log | tail -n 1 > gpurun_out/bench.json
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ]; then
  rm -rf gpurun_out/prof
  stage rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
  find gpurun_out/prof -name '*kernel_stats*' | head -n 3 | tee -a gpurun_out/run.log
fi
echo DONE | tee -a gpurun_out/run.log
