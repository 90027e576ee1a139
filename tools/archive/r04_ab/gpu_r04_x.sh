#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/x; mkdir -p $O; export TMPDIR=/tmp
{
echo "== timeline, barriers only (NRLDPC_HIP_EARLY_BN=0)"
NRLDPC_HIP_EARLY_BN=0 NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_timing.so timeout 300 python tools/task_timing.py
echo "== timeline, early bit-node start (w = waiting for the task's columns)"
NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_timing.so timeout 300 python tools/task_timing.py
} 2>&1 | grep -v amdgpu.ids | tee $O/timeline_early_bn.txt
