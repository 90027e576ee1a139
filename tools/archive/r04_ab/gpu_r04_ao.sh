#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ao; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== check-node loop split in two: the paired degree-19 tasks (the first tickets) in a loop of their own, the dispatch on the degree in the second (split) vs one loop with a branch (head = committed)"
timeout 1500 python tools/ab_fast.py head:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so split:NRLDPC_HIP_LIB=$A/libldpc_hip_split.so head2:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so split2:NRLDPC_HIP_LIB=$A/libldpc_hip_split.so
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab32_split_check_node_loop.txt
