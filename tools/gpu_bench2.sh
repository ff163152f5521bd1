#!/bin/bash
# bench at N = 1, the 2-rank gloo rehearsal of the N > 1 legs (plumbing only), config #5 phase trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 420 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; rc=$?
grep -E '^\{' gpurun_out/bench.log | tail -n 1 > gpurun_out/bench.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "avg_launch_ms")}, "clock", d["peaks_used"].get("in_kernel_clock_mhz"))
print("affinity", {k: d["affinity"][k] for k in ("ms_total", "ms_coarse_mfma", "total_over_coarse", "rows_rescanned")}, d["affinity"]["roofline"]["frac"])
c4 = d["affinity"]["config4_shard_shape"]; print("config4 shape", {k: c4[k] for k in ("ms_total", "ms_coarse_mfma", "total_over_coarse")}, c4["roofline"]["frac"])
print("cpu_baseline", d.get("cpu_baseline")); print("parity", d.get("parity"))
print("kernels", {k: v["ms"] for k, v in d["kernels"].items()})
PY
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "ABORT bench hung"; exit $rc; fi
SDK_BENCH_BACKEND=gloo timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-affinity-config3 > gpurun_out/bench_gloo2.log 2>&1; rc=$?
grep -E '^\{' gpurun_out/bench_gloo2.log | tail -n 1 > gpurun_out/bench_gloo2.json
python3 - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_gloo2.json"))
    print("gloo rehearsal: value", d["value"], "exchange", d["embedding_exchange"]["config4_shard"], "cfg4", d["config4_multi_gpu"], "cfg5", d["config5_multi_gpu"])
except Exception as e:
    print("gloo rehearsal failed", e); print(open("gpurun_out/bench_gloo2.log").read()[-3000:])
PY
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "ABORT rehearsal hung"; exit $rc; fi
timeout -k 10 300 python tools/cluster_bench.py > gpurun_out/cluster_bench.log 2>&1; tail -n 3 gpurun_out/cluster_bench.log
