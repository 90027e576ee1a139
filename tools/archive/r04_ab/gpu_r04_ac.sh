#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ac; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== ticket lookahead: the next ticket is requested at the start of a task and looked at when the task is done.  q = committed kernel; cnla / bnla / la = lookahead in the check-node loop / bit-node loop / both"
timeout 1200 python tools/ab_fast.py q: cnla:NRLDPC_HIP_LIB=$A/libldpc_hip_cnla.so bnla:NRLDPC_HIP_LIB=$A/libldpc_hip_bnla.so la:NRLDPC_HIP_LIB=$A/libldpc_hip_la.so q2: la2:NRLDPC_HIP_LIB=$A/libldpc_hip_la.so
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab24_ticket_lookahead.txt
