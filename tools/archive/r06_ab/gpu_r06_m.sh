#!/bin/bash
# round 6, call M: the UL slot (fused segment kernel) -- priority turns on/off, fresh per-workgroup trace
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 0 1 0 1; do echo "TB_FAIR=$f: $(NRLDPC_HIP_TB_FAIR=$f python tools/slot_chain.py 2>/dev/null | tail -1)"; done | tee gpurun_out/r06/slot_chain_fair.txt
for f in 0 1; do echo "TB_FAIR=$f retx: $(NRLDPC_HIP_TB_FAIR=$f python tools/slot_chain.py 20 0.18 1 2>/dev/null | tail -1)"; done | tee -a gpurun_out/r06/slot_chain_fair.txt
NRLDPC_HIP_TB_TRACE=/tmp/wg.bin python tools/slot_chain.py 5 > /dev/null 2>&1; python tools/wg_trace.py /tmp/wg.bin > gpurun_out/r06/wg_trace_fused.txt 2>&1; cp /tmp/wg.bin gpurun_out/r06/wg_fused.bin
NRLDPC_HIP_TB_FAIR=1 NRLDPC_HIP_TB_TRACE=/tmp/wg1.bin python tools/slot_chain.py 5 > /dev/null 2>&1; python tools/wg_trace.py /tmp/wg1.bin > gpurun_out/r06/wg_trace_fused_fair.txt 2>&1
cat gpurun_out/r06/wg_trace_fused.txt
