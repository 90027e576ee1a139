#!/bin/bash
# round 6, call P: double check-node tasks up to row degree 5 (HEAD) / 6 / 7
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$GRAFT_REPO_ROOT/tools/ab
timeout 900 python tools/ab_fast.py d5: d6:NRLDPC_HIP_LIB=$A/libldpc_hip_dbl6.so,NRLDPC_HIP_CN_DOUBLE=6 d7:NRLDPC_HIP_LIB=$A/libldpc_hip_dbl7.so,NRLDPC_HIP_CN_DOUBLE=7 d5b: d6b:NRLDPC_HIP_LIB=$A/libldpc_hip_dbl6.so,NRLDPC_HIP_CN_DOUBLE=6 d7b:NRLDPC_HIP_LIB=$A/libldpc_hip_dbl7.so,NRLDPC_HIP_CN_DOUBLE=7 > gpurun_out/r06/ab_double_degree.txt 2>&1
cat gpurun_out/r06/ab_double_degree.txt
for l in "" $A/libldpc_hip_dbl6.so; do NRLDPC_HIP_LIB=$l python tools/slot_chain.py 2>/dev/null | tail -1; done
