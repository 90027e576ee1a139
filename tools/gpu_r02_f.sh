#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.txt
run() { echo "$1 T=$2 case=${4:-mix}: $(env $1 timeout 120 ./tests/abi_threads.bin $L $2 ${3:-400} $4 2>&1 | tail -1 | cut -c1-330)"; }
{ run X=1 1 3000 1; run X=1 1 2000 0; run X=1 1 3000 4; run X=1 1 1200; run X=1 16 600; run X=1 32 600; run X=1 64 300; } | tee $O/abi_threads_quick.txt
timeout 300 python tests/ldpctest_hip.py -l 8448 -s 10 -n 200 2>&1 | tail -4
