#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ab; mkdir -p $O; export TMPDIR=/tmp
{
echo "== phase switch: head = committed kernel; q = one queue per pass + ticket record in one load; prep = q + head of the first bit-node ticket fetched in front of the barrier"
timeout 900 python tools/ab_fast.py head:NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_head.so q:NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_q.so prep: head2:NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_head.so q2:NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_q.so prep2:
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab23_phase_switch.txt
