#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/c; mkdir -p $O; export TMPDIR=/tmp
for S in 0 8; do
  NRLDPC_HIP_TB_STAGGER_US=$S NRLDPC_HIP_TB_TRACE=$O/trace_s$S.bin timeout 300 python tools/slot_chain.py 3 > /dev/null 2>&1
  python tools/wg_trace.py $O/trace_s$S.bin | tee $O/wg_trace_s$S.txt
done
NRLDPC_HIP_TB_STAGGER_US=0 NRLDPC_HIP_TB_TRACE=$O/trace_retx.bin timeout 300 python tools/slot_chain.py 3 0.18 retx > /dev/null 2>&1
python tools/wg_trace.py $O/trace_retx.bin | tee $O/wg_trace_retx.txt
