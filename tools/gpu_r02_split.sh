#!/bin/bash
# resident server with several CUs per block (NRLDPC_HIP_SRV_SPLIT): parity through the oracle-checked tests, latency, soak
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
for G in ${SPLITS:-2 4}; do
  S=$(NRLDPC_HIP_SRV_SPLIT=$G timeout 25 ./tests/abi_threads.bin $L 1 200 1 2>&1 | tail -1)
  US=$(echo "$S" | sed 's/.*"us_per_call_per_thread": \([0-9]*\).*/\1/')
  if [ -z "$US" ] || [ "$US" -gt 500 ]; then echo "ABORT G=$G: slow or failed: $S"; exit 1; fi
done
run() { echo "$1 T=$2 case=${4:-mix}: $(env $1 NRLDPC_HIP_SRV_DEBUG=1 timeout 40 ./tests/abi_threads.bin $L $2 ${3:-400} $4 2>&1 | grep "prologue\|calls_per_s\|CUs per block" | sed 's/.*per call (only meaningful when every call used it)//' | sed 's/"failures"/\n"failures"/' | cut -c1-250 | tr '\n' ' ')"; }
{
for G in ${SPLITS:-2 4}; do
  run NRLDPC_HIP_SRV_SPLIT=$G 1 3000 1; run NRLDPC_HIP_SRV_SPLIT=$G 1 2000 0; run NRLDPC_HIP_SRV_SPLIT=$G 1 1200; run NRLDPC_HIP_SRV_SPLIT=$G 16 600
done
} | tee $O/abi_threads_split.txt
if [ -n "$QUICK" ]; then exit 0; fi
for G in ${SPLITS:-2 4}; do
  NRLDPC_HIP_SRV_SPLIT=$G timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q -k "per_segment or concurrent or ldpctest or reference_entry" 2>&1 | tail -3
  NRLDPC_HIP_SRV_SPLIT=$G timeout 100 python tests/ldpctest_hip.py -l 8448 -s 10 -n 300 2>&1 | tail -2
done
srun() { echo "soak $1 T=$2: $(env $1 timeout 60 ./tests/abi_threads.bin $L $2 $3 2>&1 | tail -1 | cut -c1-120)"; }
for G in ${SPLITS:-2 4}; do srun NRLDPC_HIP_SRV_SPLIT=$G 8 6000; srun NRLDPC_HIP_SRV_SPLIT=$G 1 10000; srun NRLDPC_HIP_SRV_SPLIT=$G 32 2000; done
