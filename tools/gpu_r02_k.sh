#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
timeout 1500 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 40 --no-cpu-baseline --no-strong 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench ms', d['ms_per_step'], 'value', d['value'], 'op', d.get('operating_point'))"
run() { echo "$1 T=$2 case=${4:-mix}: $(env $1 NRLDPC_HIP_SRV_DEBUG=1 timeout 25 ./tests/abi_threads.bin $L $2 ${3:-400} $4 2>&1 | grep "prologue\|calls_per_s" | sed 's/.*per call (only meaningful when every call used it)//' | cut -c1-330 | tr '\n' ' ')"; }
{ run X=1 1 3000 1; run X=1 1 2000 0; run X=1 1 1200; run X=1 32 600; } | tee $O/abi_threads_quick.txt
python tools/slot_chain.py 30
