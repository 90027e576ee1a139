#!/bin/bash
# round 6, call F: per-block timelines, one workgroup per block vs persistent workgroups; headline code and a two-workgroups-per-CU code
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for code in 1,384,13 1,384,23; do for p in 0 1; do
  DEC_TRACE_CODE=$code NRLDPC_HIP_PERSIST=$p timeout 300 python tools/dec_trace.py > gpurun_out/r06/dec_trace_${code//,/_}_persist$p.txt 2>&1
done; done
tail -n 30 gpurun_out/r06/dec_trace_1_384_13_persist1.txt
