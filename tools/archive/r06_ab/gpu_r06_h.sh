#!/bin/bash
# round 6, call H: co-resident workgroups taking turns at the issue priority (NRLDPC_HIP_FAIR) -- A/B on every code + a trace
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NRLDPC_HIP_PERSIST=0 timeout 900 python tools/ab_fast.py f0:NRLDPC_HIP_FAIR=0 f1:NRLDPC_HIP_FAIR=1 t8:NRLDPC_HIP_FAIR=8 t9:NRLDPC_HIP_FAIR=9 t10:NRLDPC_HIP_FAIR=10 f0b:NRLDPC_HIP_FAIR=0 t9b:NRLDPC_HIP_FAIR=9 > gpurun_out/r06/ab_fair.txt 2>&1
cat gpurun_out/r06/ab_fair.txt
DEC_TRACE_CODE=1,384,23 NRLDPC_HIP_PERSIST=0 NRLDPC_HIP_FAIR=9 timeout 300 python tools/dec_trace.py gpurun_out/r06/raw_r23_fair9 > gpurun_out/r06/dec_trace_1_384_23_fair.txt 2>&1
