#!/bin/bash
# res2net chain: parity tests, in-kernel timeline, short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q --no-header -p no:cacheprovider -x -k "res2net or ecapa" > gpurun_out/r2_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/r2_tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
timeout -k 10 300 python tools/res2net_timeline.py > gpurun_out/res2net_timeline.log 2>&1 || { tail gpurun_out/res2net_timeline.log; exit 1; }
grep -v amdgpu.ids gpurun_out/res2net_timeline.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-affinity-config3 > gpurun_out/bench_r2.log 2>&1 || { tail -n 20 gpurun_out/bench_r2.log; exit 1; }
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_r2.log") if l.startswith("{")][-1])
print("bench:", d["value"], "seg/s", d["ms_per_step"], "ms/step;", {k: v["ms"] for k, v in d["kernels"].items()})
PY
