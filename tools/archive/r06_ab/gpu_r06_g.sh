#!/bin/bash
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DEC_TRACE_CODE=1,384,23 NRLDPC_HIP_PERSIST=0 timeout 300 python tools/dec_trace.py gpurun_out/r06/raw_r23_p0 > /dev/null 2>&1
DEC_TRACE_CODE=1,384,23 NRLDPC_HIP_PERSIST=1 timeout 300 python tools/dec_trace.py gpurun_out/r06/raw_r23_p1 > /dev/null 2>&1
ls -la gpurun_out/r06/
