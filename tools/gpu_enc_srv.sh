#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
timeout 120 python tools/enc_call_latency.py 2000 2>&1 | grep -v amdgpu | sed "s/^/server: /"
NRLDPC_HIP_ENC_SERVER=0 timeout 120 python tools/enc_call_latency.py 1000 2>&1 | grep -v amdgpu | sed "s/^/launch per call: /"
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_decoder.py -m gpu -x -q 2>&1 | tail -3
timeout 60 ./tests/abi_threads.bin $L 1 3000 1 | tail -1 | cut -c1-250
timeout 300 python tests/ldpctest_hip.py -l 8448 -s 10 -n 100 2>&1 | tail -2
