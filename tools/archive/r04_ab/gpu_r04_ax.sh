#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ax; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== persistent form of the homogeneous launch (one workgroup per slot of the GPU, blocks taken with stride gridDim; tables copied into LDS once per workgroup): NRLDPC_HIP_PERSIST=1 vs 0; head = committed library"
timeout 1500 python tools/ab_fast.py head:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so p0: p1:NRLDPC_HIP_PERSIST=1 p0b: p1b:NRLDPC_HIP_PERSIST=1
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab36_persistent_workgroups.txt
