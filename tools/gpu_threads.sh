#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
run() { echo "$1 T=$2: $(env $1 timeout 120 ./tests/abi_threads.bin $L $2 ${3:-400} 2>&1 | tail -1 | cut -c1-110)"; }
for T in ${THREADS:-1 4 16 32 64}; do
  run NRLDPC_HIP_AGGREGATE=0 $T
  for F in ${INFLIGHT:-1 2 4 8}; do run "NRLDPC_HIP_AGGREGATE=1 NRLDPC_HIP_AGG_INFLIGHT=$F" $T; done
done | tee gpurun_out/abi_threads.txt
