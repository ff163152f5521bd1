#!/bin/bash
# N > 1 plumbing rehearsal on the one-GPU box: 4 ranks share the device (the box admits at most 6 processes with the GPU open; 6 ranks + the launcher were refused), collectives staged through gloo.
# Not a measurement: it shows that `python bench.py --gpus N` finishes inside the driver's 600 s, fits, and fills every N > 1 object of the line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
t0=$(date +%s)
SDK_BENCH_BACKEND=gloo timeout -k 10 580 python bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r5_bench_gloo4.json 2> gpurun_out/r5_bench_gloo4.err; rc=$?
echo "rc $rc, $(( $(date +%s) - t0 )) s"; tail -n 5 gpurun_out/r5_bench_gloo4.err
python - <<'P'
import json
l=[x for x in open('gpurun_out/r5_bench_gloo4.json') if x.startswith('{')]
j=json.loads(l[-1])
print({k: j[k] for k in ('value','n_gpus','ms_per_step')}, j['embedding_exchange'], j['config4_multi_gpu'], j['config5_multi_gpu'], j.get('collective_env'))
P
