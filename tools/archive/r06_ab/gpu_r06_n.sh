#!/bin/bash
# round 6, call N: r04's double check-node tasks (two low-degree row items per thread) on the build without scratch memory
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$GRAFT_REPO_ROOT/tools/ab
export NRLDPC_HIP_PERSIST=0
timeout 900 python tools/ab_fast.py head:NRLDPC_HIP_LIB=$A/libldpc_hip_r06a.so d0:NRLDPC_HIP_CN_DOUBLE=0 d5:NRLDPC_HIP_CN_DOUBLE=5 d4:NRLDPC_HIP_CN_DOUBLE=4 headb:NRLDPC_HIP_LIB=$A/libldpc_hip_r06a.so d5b:NRLDPC_HIP_CN_DOUBLE=5 > gpurun_out/r06/ab_double_tasks.txt 2>&1
cat gpurun_out/r06/ab_double_tasks.txt
