#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
run() { echo "=== $1 T=$2 N=$3 case=$4"; env $1 timeout 60 ./tests/abi_threads.bin $L $2 $3 $4 2>&1 | tail -${5:-3} | cut -c1-420; }
{
  for i in 1 2 3; do run "NRLDPC_HIP_SRV_SLOTS=4" 1 3000 1; done
  run NRLDPC_HIP_SRV_DEBUG=1 1 3000 1 2
  run NRLDPC_HIP_SRV_DEBUG=1 1 3000 0 2
  run NRLDPC_HIP_SRV_DEBUG=1 1 3000 4 2
  run NRLDPC_HIP_SRV_DEBUG=1 1 3000 6 2
  for i in 1 2 3; do run X=1 32 600 ""; done
  for i in 1 2; do run X=1 64 300 ""; done
  run X=1 16 600 ""
  run X=1 4 600 ""
} > $O/dbg.txt 2>&1
grep -c "Memory access fault" $O/dbg.txt
cat $O/dbg.txt
