#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
run() { echo "T=$1 case=${3:-mix}: $(timeout 120 ./tests/abi_threads.bin $L $1 $2 $3 2>&1 | tail -2 | cut -c1-230)"; }
for i in 1 2 3 4 5 6; do run 8 6000; done
for i in 1 2 3 4; do run 1 10000; done
run 32 3000
run 64 1500
timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q -k "per_segment or concurrent or ldpctest" 2>&1 | tail -3
