#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
run() { echo "$1 T=$2 case=${4:-mix}: $(env $1 NRLDPC_HIP_SRV_DEBUG=1 timeout 25 ./tests/abi_threads.bin $L $2 ${3:-400} $4 2>&1 | grep "prologue\|calls_per_s" | sed 's/.*per call (only meaningful when every call used it)//' | cut -c1-330 | tr '\n' ' ')"; }
# sanity first: a broken completion path makes every call wait for the idle timeout
S=$(timeout 25 ./tests/abi_threads.bin $L 1 200 1 2>&1 | tail -1)
echo "sanity: $S" | cut -c1-300
US=$(echo "$S" | sed 's/.*"us_per_call_per_thread": \([0-9]*\).*/\1/')
if [ -z "$US" ] || [ "$US" -gt 500 ]; then echo "ABORT: slow or failed"; exit 1; fi
{ run X=1 1 3000 1; run X=1 1 2000 0; run X=1 1 1200; run X=1 32 600; } | tee $O/abi_threads_quick.txt
srun() { echo "soak T=$1 case=${3:-mix}: $(timeout 60 ./tests/abi_threads.bin $L $1 $2 $3 2>&1 | tail -1 | cut -c1-120)"; }
for i in 1 2 3; do srun 8 6000; done
for i in 1 2; do srun 1 10000; done
srun 32 3000
srun 64 1500
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q -k "per_segment or concurrent or ldpctest or reference_entry" 2>&1 | tail -3
