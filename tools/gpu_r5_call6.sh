#!/bin/bash
# round 5, call 6 (closing validation of the final tree): the whole suite, smoke, the bench line as the driver runs it, rocprofv3 kernel statistics of the same command, the 4-rank rehearsal
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tests_6.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r5_tests_6.log | tail -n 12
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r5_smoke.log 2>&1 || { tail gpurun_out/r5_smoke.log; exit 1; }
tail -n 2 gpurun_out/r5_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/r5_bench_6.json 2> gpurun_out/r5_bench_6.err || { tail -n 20 gpurun_out/r5_bench_6.err; exit 1; }
python - <<'P'
import json
j = json.loads([l for l in open('gpurun_out/r5_bench_6.json') if l.startswith('{')][-1])
pm = j['precision_modes']
print({k: j[k] for k in ('value', 'ms_per_step', 'value_at_north_star_tolerance', 'value_from_host')}, 'frac', j['roofline']['frac'], 'traffic', j['roofline']['traffic'])
print('fp16', pm['fp16']['value'], pm['fp16']['max_abs_dscore_all_pairs'], 'aff', j['affinity']['ms_total'], j['affinity']['roofline']['frac'], j['affinity']['roofline']['traffic'], 'xv', j['xvector']['value'])
P
rm -rf gpurun_out/prof6
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6 -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r5_rocprof.log 2>&1 || { tail gpurun_out/r5_rocprof.log; exit 1; }
find gpurun_out/prof6 -name '*kernel_stats*' | head -n 2
bash tools/gpu_r5_gloo4.sh
timeout -k 10 120 python tools/sustained.py > gpurun_out/r5_sustained.json 2> gpurun_out/r5_sustained.err || { tail gpurun_out/r5_sustained.err; exit 1; }
tail -n 1 gpurun_out/r5_sustained.json
timeout -k 10 300 python tools/four_procs.py 4 > gpurun_out/r5_four_procs.json 2> gpurun_out/r5_four_procs.err || { tail gpurun_out/r5_four_procs.err; exit 1; }
tail -n 1 gpurun_out/r5_four_procs.json
