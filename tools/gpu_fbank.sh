#!/bin/bash
# fbank: parity tests, then a short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backend_e2e.py tests/test_gpu_parity_fp32.py -q --no-header -p no:cacheprovider -x -k "fbank or e2e or parity or ecapa" > gpurun_out/fb_tests.log 2>&1; rc=$?
tail -n 5 gpurun_out/fb_tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-affinity-config3 > gpurun_out/bench_fb.log 2>&1 || { tail -n 20 gpurun_out/bench_fb.log; exit 1; }
python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_fb.log") if l.startswith("{")][-1])
print("bench:", d["value"], "seg/s", d["ms_per_step"], "ms/step;", {k: v["ms"] for k, v in d["kernels"].items()})
PY
