#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/an; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== double check-node tasks, the two-item bodies as a called (not inlined) function: head = committed; n0 / n5 = that build with NRLDPC_HIP_CN_DOUBLE=0 / default"
timeout 1500 python tools/ab_fast.py head:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so n0:NRLDPC_HIP_LIB=$A/libldpc_hip_cn2ni.so,NRLDPC_HIP_CN_DOUBLE=0 n5:NRLDPC_HIP_LIB=$A/libldpc_hip_cn2ni.so head2:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so n5b:NRLDPC_HIP_LIB=$A/libldpc_hip_cn2ni.so
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab31_double_tasks_called_function.txt
