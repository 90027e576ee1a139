#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/l; mkdir -p $O; export TMPDIR=/tmp
NRLDPC_HIP_TB_TRACE=$O/trace.bin timeout 300 python tools/slot_chain.py 3 > /dev/null 2>&1; python tools/wg_trace.py $O/trace.bin | tee $O/wg_trace.txt; rm -f $O/trace.bin
NRLDPC_HIP_TB_TRACE=$O/trace.bin timeout 300 python tools/slot_chain.py 3 0.18 retx > /dev/null 2>&1; python tools/wg_trace.py $O/trace.bin | tee $O/wg_trace_retx.txt; rm -f $O/trace.bin
