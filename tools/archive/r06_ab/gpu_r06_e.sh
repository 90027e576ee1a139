#!/bin/bash
# round 6, call E: parity-stop instantiation without the CRC check (91 VGPRs, no scratch) vs round 5's library, all codes
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$GRAFT_REPO_ROOT/tools/ab
NRLDPC_HIP_PERSIST=0 timeout 900 python tools/ab_fast.py r05:NRLDPC_HIP_LIB=$A/libldpc_hip_r05.so new: r05b:NRLDPC_HIP_LIB=$A/libldpc_hip_r05.so newb: > gpurun_out/r06/ab_no_scratch.txt 2>&1
cat gpurun_out/r06/ab_no_scratch.txt
