#!/bin/bash
# round 5, call 1: the whole GPU suite, the bench line, the start-up clock probe and the ingest copy trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tests_1.log 2>&1; rc=$?
tail -n 15 gpurun_out/r5_tests_1.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r5_bench_1.json 2> gpurun_out/r5_bench_1.err || { tail -n 20 gpurun_out/r5_bench_1.err; exit 1; }
python - <<'P'
import json
j = json.loads([l for l in open('gpurun_out/r5_bench_1.json') if l.startswith('{')][-1])
print({k: j[k] for k in ('value', 'ms_per_step', 'value_at_north_star_tolerance', 'value_from_host')}, j['roofline']['frac'], j['xvector']['value'], j['xvector']['timing'], j['ingest']['ratio_to_resident_same_leg'], j['ingest']['limiter'])
P
timeout -k 10 120 python tools/startup_clock.py > gpurun_out/r5_startup_clock.json 2> gpurun_out/r5_startup_clock.err || { tail gpurun_out/r5_startup_clock.err; exit 1; }
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r5_ingest_trace" -o ingest -- python3 "$GRAFT_REPO_ROOT/tools/one_ingest.py" > "$GRAFT_REPO_ROOT/gpurun_out/r5_ingest_trace.log" 2>&1
echo "rocprof rc $?"; ls "$GRAFT_REPO_ROOT/gpurun_out/r5_ingest_trace" | head
