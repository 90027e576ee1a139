#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/am; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== double check-node tasks against the committed kernel on one box: head = committed; d0 / d5 = the new build with NRLDPC_HIP_CN_DOUBLE=0 / default"
timeout 1500 python tools/ab_fast.py head:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so d0:NRLDPC_HIP_CN_DOUBLE=0 d5: head2:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so d5b:
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab31_double_check_node_tasks_vs_head.txt
