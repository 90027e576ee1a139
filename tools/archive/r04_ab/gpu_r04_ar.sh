#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ar; mkdir -p $O; export TMPDIR=/tmp
for T in 0 1; do
NRLDPC_HIP_TB_W6=$T NRLDPC_HIP_TB_TRACE=$O/trace.bin timeout 300 python tools/slot_chain.py 3 > /dev/null 2>&1; python tools/wg_trace.py $O/trace.bin > $O/wg_trace_w6_$T.txt; rm -f $O/trace.bin
done
head -40 $O/wg_trace_w6_1.txt
