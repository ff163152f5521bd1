#!/bin/bash
# round 5, call 4: the whole GPU suite on the tree with precision 2, the bench line (fp16 object), then the 4-rank gloo rehearsal of --gpus N
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tests_4.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r5_tests_4.log | tail -n 15
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r5_bench_4.json 2> gpurun_out/r5_bench_4.err || { tail -n 20 gpurun_out/r5_bench_4.err; exit 1; }
python - <<'P'
import json
j = json.loads([l for l in open('gpurun_out/r5_bench_4.json') if l.startswith('{')][-1])
pm = j['precision_modes']
print({k: j[k] for k in ('value', 'ms_per_step', 'value_at_north_star_tolerance', 'value_from_host')}, j['roofline']['frac'])
print('fp16', pm['fp16'])
print('precise', pm['precise']['value'], pm['precise']['max_abs_dscore_all_pairs'], 'default dscore', pm['default']['max_abs_dscore_all_pairs'])
P
bash tools/gpu_r5_gloo4.sh
