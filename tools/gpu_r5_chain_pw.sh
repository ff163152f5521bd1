#!/bin/bash
# sustained GEMM time + board power per BINARY (SDK_HIP_LIB): LIBS="a.so b.so" [MODES="gemm gemm_k1024"] [REPS=2]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
for lib in ${LIBS:-tools/probe/libsdk_hip_r5final.so speaker-diarization-toolkit_amd/libsdk_hip.so}; do
  echo "== $lib" | tee -a gpurun_out/r5_chain_power.txt
  SDK_HIP_LIB=$PWD/$lib timeout -k 10 200 python tools/power_probe.py ${MODES:-gemm gemm_k1024} 2>/dev/null | tail -n 1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
for p in j['phases']:
    if p['mode'] != 'idle': print(p['mode'], p['power_mean_W'], 'W', p['sclk_mean_MHz'], 'MHz', p['ms_per_call'], 'ms')
" | tee -a gpurun_out/r5_chain_power.txt || exit 1
done
done
