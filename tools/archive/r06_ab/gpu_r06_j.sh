#!/bin/bash
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in 0 1; do
  DEC_TRACE_CODE=2,384,15 NRLDPC_HIP_PERSIST=$p timeout 300 python tools/dec_trace.py > gpurun_out/r06/dec_trace_2_384_15_persist$p.txt 2>&1
done
grep -h "kernel span\|workgroup: total\|n_iter = \|gap between" gpurun_out/r06/dec_trace_2_384_15_persist*.txt | cut -c1-330
